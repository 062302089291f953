"""Per-kernel, per-launch means of the counters collected by tools/pmc_profile.sh, plus a few derived ratios
(tools/make_traffic_json.py turns the same files into per-frame totals)."""
import collections
import csv
import glob
import json
import sys

out = collections.defaultdict(dict)
for f in sorted(glob.glob(sys.argv[1] + "/pass*/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" not in k:
            continue
        k = k[k.index("k_"):].split("(")[0]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
for k, c in sorted(out.items()):
    d = dict(c)
    g = d.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        d["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        # vector instructions issued over what the SIMDs can issue in the kernel's active cycles (one wave64 instruction per
        # SIMD per 2 cycles).  GRBM_GUI_ACTIVE arrives summed over the eight XCDs, each with 128 SIMDs; SQ_BUSY_CYCLES is
        # summed over shader engines and is no denominator for this
        d["valu_issue_frac"] = g("SQ_INSTS_VALU") / (128 * g("GRBM_GUI_ACTIVE") / 2.0)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        d["l1_hit_rate_est"] = 1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["hbm_bytes_per_launch"] = 2 * 1024 * g("FETCH_SIZE") + 1024 * g("WRITE_SIZE")   # gfx950: FETCH_SIZE counts half
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    print(k)
    print(json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in sorted(d.items())}, indent=1))
