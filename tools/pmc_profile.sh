#!/bin/bash
# Hardware-counter passes over the bench (one rocprofv3 run per pass; counters are never mixed with tracing).
# usage: tools/pmc_profile.sh <out_dir_under_gpurun_out> [workload]
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/$1"; WL="${2:-bunny_lambert_1080p}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
case $WL in
  bunny_field_4k) LIMIT=600;;
  mirror_room|glass_balls|bunny_phong_1080p) LIMIT=240;;
  *) LIMIT=120;;
esac
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES"
)
i=0
for p in "${PASSES[@]}"; do
  timeout -s KILL $LIMIT rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -o p -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --no-roofline --workload "$WL" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
  i=$((i+1))
done
python3 "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"; cat "$OUT/summary.txt"
