#!/bin/bash
# Everything bench.py's roofline blocks quote, collected in one go on the GPU box:
#   rocprofv3 --kernel-trace --stats of the bench, then the hardware-counter passes (one rocprofv3 --pmc run per pass),
#   for each workload given.  usage: tools/collect_profiles.sh <tag> [workload ...]   -> gpurun_out/<tag>/
# Afterwards, in the build container: python tools/make_traffic_json.py <tag>  (writes profiles/traffic.json).
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; shift
WLS="${@:-bunny_lambert_1080p bunny_lambert_2lights_1080p}"
mkdir -p "$R/gpurun_out/$TAG"; cd /tmp; export TMPDIR=/tmp
for w in $WLS; do
  case $w in
    bunny_field_4k) STEPS="--steps 2 --warmup 1"; LIMIT=500;;
    mirror_room|glass_balls|bunny_phong_1080p) STEPS="--steps 5 --warmup 1"; LIMIT=300;;
    *) STEPS="--steps 20 --warmup 3"; LIMIT=200;;
  esac
  timeout -s KILL $LIMIT rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$TAG/kt_$w" -o kt -- python3 "$R/bench.py" $STEPS --no-cpu-baseline --no-workloads --no-roofline --workload $w > "$R/gpurun_out/$TAG/kt_$w.log" 2>&1 || { echo "kernel trace of $w failed"; tail -5 "$R/gpurun_out/$TAG/kt_$w.log"; exit 1; }
  find "$R/gpurun_out/$TAG/kt_$w" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/$TAG/${w}_kernel_stats.csv" \;
  rm -f "$R/gpurun_out/$TAG/kt_$w"/*kernel_trace.csv "$R/gpurun_out/$TAG/kt_$w"/*/*kernel_trace.csv 2>/dev/null
  echo "$w: kernel trace done"
  "$R/tools/pmc_profile.sh" "$TAG/pmc_$w" $w > "$R/gpurun_out/$TAG/${w}_pmc_summary.txt" 2>&1 || { echo "counter passes of $w failed"; tail -5 "$R/gpurun_out/$TAG/${w}_pmc_summary.txt"; exit 1; }
  echo "$w done"
done
