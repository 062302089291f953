#!/bin/bash
# rocprofv3 kernel trace of one bench run per workload; prints total time per kernel.  usage: tools/kernel_breakdown.sh <tag> <workload ...>
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; shift
mkdir -p "$R/gpurun_out/$TAG"; cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$TAG/kt_$w" -o kt -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-workloads --workload $w > "$R/gpurun_out/$TAG/kt_$w.log" 2>&1 || { echo "kernel trace of $w failed"; tail -5 "$R/gpurun_out/$TAG/kt_$w.log"; exit 1; }
  find "$R/gpurun_out/$TAG/kt_$w" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/$TAG/${w}_kernel_stats.csv" \;
  echo "== $w"; python3 - "$R/gpurun_out/$TAG/${w}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:10]:
    n = r["Name"]; k = n[n.index("k_"):].split("(")[0] if "k_" in n else n[:40]
    print("  %-28s calls %5s  total %9.3f ms  %5.1f %%  avg %9.1f us" % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
PY
done
