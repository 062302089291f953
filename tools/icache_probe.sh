#!/bin/bash
# Instruction-cache and scalar-data-cache counters of the kernels (one rocprofv3 --pmc run per workload):
# usage: tools/icache_probe.sh <tag> <workload ...>
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; shift
mkdir -p "$R/gpurun_out/$TAG"; cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d "$R/gpurun_out/$TAG/ic_$w" -o p -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --no-roofline --workload $w > "$R/gpurun_out/$TAG/ic_$w.log" 2>&1 || { echo "$w failed"; tail -5 "$R/gpurun_out/$TAG/ic_$w.log"; exit 1; }
  echo "== $w"; python3 - "$R/gpurun_out/$TAG/ic_$w" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" not in k: continue
        k = k[k.index("k_"):].split("(")[0]
        tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
for k in sorted({a for a, _ in tot}):
    g = lambda c: tot.get((k, c), 0.0)
    if g("SQC_ICACHE_REQ"):
        print("  %-26s icache miss rate %.4f (%.3g misses)  scalar dcache miss rate %.4f  ifetch per wave-cycle %.4f" % (
            k, g("SQC_ICACHE_MISSES") / g("SQC_ICACHE_REQ"), g("SQC_ICACHE_MISSES"), g("SQC_DCACHE_MISSES") / max(g("SQC_DCACHE_REQ"), 1), g("SQ_IFETCH") / max(g("SQ_WAVE_CYCLES"), 1)))
PY
done
