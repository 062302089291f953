#!/bin/bash
# Everything bench.py's roofline block quotes, collected in one go on the GPU box:
#   rocprofv3 --kernel-trace --stats of the bench, then the hardware-counter passes (one rocprofv3 --pmc run per pass),
#   for each workload given.  usage: tools/collect_profiles.sh <tag> [workload ...]   -> gpurun_out/<tag>/
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; shift
WLS="${@:-bunny_lambert_1080p bunny_lambert_2lights_1080p}"
mkdir -p "$R/gpurun_out/$TAG"; cd /tmp; export TMPDIR=/tmp
for w in $WLS; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$TAG/kt_$w" -o kt -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-workloads --workload $w > "$R/gpurun_out/$TAG/kt_$w.log" 2>&1 || { echo "kernel trace of $w failed"; tail -5 "$R/gpurun_out/$TAG/kt_$w.log"; exit 1; }
  cp "$R/gpurun_out/$TAG/kt_$w"/*kernel_stats.csv "$R/gpurun_out/$TAG/${w}_kernel_stats.csv" 2>/dev/null || find "$R/gpurun_out/$TAG/kt_$w" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/$TAG/${w}_kernel_stats.csv" \;
  "$R/tools/pmc_profile.sh" "$TAG/pmc_$w" $w > "$R/gpurun_out/$TAG/${w}_pmc_summary.txt" 2>&1 || { echo "counter passes of $w failed"; tail -5 "$R/gpurun_out/$TAG/${w}_pmc_summary.txt"; exit 1; }
  echo "$w done"
done
