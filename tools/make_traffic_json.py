"""Writes profiles/traffic.json (what bench.py's roofline blocks read) from what tools/collect_profiles.sh left behind.

    python tools/make_traffic_json.py <dir under gpurun_out, e.g. r03_p> [more dirs ...]

Per workload the directory holds pmc_<workload>/pass*/p_counter_collection.csv (one rocprofv3 --pmc run per pass over
`bench.py --no-roofline`) and <workload>_kernel_stats.csv + kt_<workload>/..._kernel_trace.csv (rocprofv3 --kernel-trace
--stats of the same command).  A frame of the heavier workloads is many launches of each kernel (levels of the ray tree x
bands of the image), so everything is kept PER FRAME: a counter's values are summed over all launches of a kernel in a
run and divided by the frames of that run (every frame of the bench ends with exactly one k_resolve_u8 launch, so the
frames of a run are counted in the same file).  Launches of the counting instances of the kernels (template argument
COUNT = true, one untimed frame per run) are left out.
HBM bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE: both counters are in KiB and come from separate passes; on gfx950
FETCH_SIZE reports half the bytes read at every load width (measured with tools/ubench/fetch_calib.hip), so reads are
doubled.  The file is stamped with the hash of the kernel sources it was collected on; bench.py ignores it when that is
not the hash of the sources it runs."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def short(name):
    k = name[name.index("k_"):]
    base = k.split("<")[0].split("(")[0]
    targs = k.split("<")[1].split(">")[0].replace(" ", "").split(",") if "<" in k else []
    counting = len(targs) > 1 and targs[1] in ("true", "1")      # k_extend / k_shadow <EXACT, COUNT, ...>
    return base, counting


def counters(pmc_dir):
    """{kernel: {counter: per-frame total}}, {kernel: launches per frame}"""
    out = collections.defaultdict(dict)
    launches = {}
    for f in sorted(glob.glob(os.path.join(pmc_dir, "pass*", "p_counter_collection.csv"))):
        tot = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if "k_" not in r["Kernel_Name"]:
                continue
            base, counting = short(r["Kernel_Name"])
            if counting:
                continue
            tot[(base, r["Counter_Name"])] += float(r["Counter_Value"])
            disp[base].add(r["Dispatch_Id"])
        frames = len(disp.get("k_resolve_u8", ()))
        if not frames:
            continue
        for (k, c), v in tot.items():
            out[k][c] = v / frames
        for k, d in disp.items():
            launches[k] = round(len(d) / frames, 2)
    return out, launches


def durations(stats_csv, trace_csv):
    """{kernel: ms per frame} from the kernel trace of a bench run (frames = k_resolve_u8 launches of that run)."""
    out = {}
    if not stats_csv or not os.path.exists(stats_csv):
        return out
    tot = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(stats_csv)):
        if "k_" not in r["Name"]:
            continue
        base, counting = short(r["Name"])
        if counting:
            continue
        tot[base][0] += float(r["TotalDurationNs"]); tot[base][1] += int(r["Calls"])
    frames = tot["k_resolve_u8"][1] if "k_resolve_u8" in tot else 0
    if not frames:
        return out
    for k, (ns, calls) in tot.items():
        out[k] = {"ms_per_frame": round(ns / frames / 1e6, 4), "launches_per_frame": round(calls / frames, 2),
                  "avg_launch_ms": round(ns / calls / 1e6, 4)}
    return out


def workload_doc(d, wl):
    c, launches = counters(os.path.join(d, "pmc_" + wl))
    t = durations(os.path.join(d, wl + "_kernel_stats.csv"), None)
    wdoc = {}
    for k, v in sorted(c.items()):
        e = {"launches_per_frame": launches.get(k)}
        g = v.get
        if "SQ_INSTS_VALU" in v:
            e["valu_insts"] = int(v["SQ_INSTS_VALU"])
        if "SQ_INSTS_SALU" in v:
            e["salu_insts"] = int(v["SQ_INSTS_SALU"])
        if "SQ_INSTS_VMEM_RD" in v:
            e["vmem_rd_insts"] = int(v["SQ_INSTS_VMEM_RD"])
        if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
            e["lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 3)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["hbm_bytes"] = int(2 * 1024 * v["FETCH_SIZE"] + 1024 * v["WRITE_SIZE"])
            e["hbm_read_bytes"] = int(2 * 1024 * v["FETCH_SIZE"])
        if g("TCP_TOTAL_CACHE_ACCESSES_sum") and "TCP_TCC_READ_REQ_sum" in v:
            e["l1_hit_rate_est"] = round(1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
            e["l1_accesses"] = int(v["TCP_TOTAL_CACHE_ACCESSES_sum"])
            e["l2_read_requests"] = int(v["TCP_TCC_READ_REQ_sum"])
        if "TCC_HIT_sum" in v and v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
        if g("TD_TD_BUSY_sum") and g("SQ_BUSY_CU_CYCLES"):
            e["td_busy_frac"] = round(v["TD_TD_BUSY_sum"] / v["SQ_BUSY_CU_CYCLES"], 3)     # L1 data-return path busy
        if g("TA_TA_BUSY_sum") and g("SQ_BUSY_CU_CYCLES"):
            e["ta_busy_frac"] = round(v["TA_TA_BUSY_sum"] / v["SQ_BUSY_CU_CYCLES"], 3)     # L1 address path busy
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
            e["wave_wait_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)         # wave time spent waiting (mostly memory)
        if k in t:
            e["rocprof_ms_per_frame"] = t[k]["ms_per_frame"]
            e["rocprof_avg_launch_ms"] = t[k]["avg_launch_ms"]
            e["launches_per_frame"] = t[k]["launches_per_frame"]
        wdoc[k] = e
    return wdoc


def main():
    doc = {"kernel_src_sha": bench.kernel_source_sha(),
           "source": "tools/collect_profiles.sh (rocprofv3 --kernel-trace --stats, then one rocprofv3 --pmc run per counter pass), "
                     "summarised per frame by tools/make_traffic_json.py",
           "workloads": {}}
    for d in sys.argv[1:]:
        d = d if os.path.isdir(d) else os.path.join(ROOT, "gpurun_out", d)
        for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
            wl = os.path.basename(p)[4:]
            doc["workloads"][wl] = workload_doc(d, wl)
    path = os.path.join(ROOT, "profiles", "traffic.json")
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    for wl, w in doc["workloads"].items():
        print(wl)
        for k, e in w.items():
            print("   %-14s %s" % (k, json.dumps(e, sort_keys=True)))


if __name__ == "__main__":
    main()
