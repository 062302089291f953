#!/bin/bash
# Quick GPU check used while tuning: smoke (bit-equality with the oracle), then the bench without the CPU leg.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 200 python -u __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
for w in ${WORKLOADS:-bunny_lambert_1080p}; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-workloads --workload $w ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['workload'], 'ms/frame', d['ms_per_step'], 'Mrays/s', d['value'], '| extend', r['ms_extend'], 'shadow', r['ms_shadow'], 'replay', r['ms_replay'], '|', r['kernel'], 'nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'], 'GB/s', r['achieved'])
" || exit 1
done
