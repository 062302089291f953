"""Writes profiles/traffic.json (what bench.py's roofline block reads) from hardware-counter passes and a kernel trace.

    python tools/make_traffic_json.py <workload>=<pmc_dir>[,<kernel_stats.csv>] ...

<pmc_dir> is what tools/pmc_profile.sh wrote (pass*/p_counter_collection.csv, one rocprofv3 --pmc run per pass);
<kernel_stats.csv> the `rocprofv3 --kernel-trace --stats` summary of the same command.  The file is stamped with the hash
of the kernel sources it was collected on; bench.py ignores it when that is not the hash of the sources it runs.
Counter values are means over the launches of the plain (non-counting) kernels.  HBM bytes = 2 * 1024 * FETCH_SIZE +
1024 * WRITE_SIZE: both counters are in KiB and come from separate passes; on gfx950 FETCH_SIZE reports half the bytes
read, at every load width (measured with tools/ubench/fetch_calib.hip: 16-, 4- and 1-byte loads per lane), so reads are
doubled."""
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
    counting = "<" in k and k.split("<")[1].split(">")[0].replace(" ", "").endswith(",true")
    return base, counting


def counters(pmc_dir):
    out = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(pmc_dir, "pass*", "p_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_" not in r["Kernel_Name"]:
                continue
            base, counting = short(r["Kernel_Name"])
            if counting:
                continue
            agg[(base, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            out[k][c] = sum(v) / len(v)
    return out


def durations(stats_csv):
    out = {}
    if not stats_csv:
        return out
    tot = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(stats_csv)):
        if "k_" not in r["Name"]:
            continue
        base, counting = short(r["Name"])
        if counting:
            continue
        tot[base][0] += float(r["TotalDurationNs"]); tot[base][1] += int(r["Calls"])
    for k, (ns, calls) in tot.items():
        out[k] = round(ns / calls / 1e6, 4)
    return out


def main():
    doc = {"kernel_src_sha": bench.kernel_source_sha(),
           "source": "tools/pmc_profile.sh passes + rocprofv3 --kernel-trace --stats, summarised by tools/make_traffic_json.py",
           "workloads": {}}
    for arg in sys.argv[1:]:
        wl, rest = arg.split("=", 1)
        pmc_dir, _, stats_csv = rest.partition(",")
        c, d = counters(pmc_dir), durations(stats_csv)
        wdoc = {}
        for k, v in sorted(c.items()):
            e = {}
            if "SQ_INSTS_VALU" in v:
                e["valu_insts"] = int(v["SQ_INSTS_VALU"])
            if "SQ_INSTS_SALU" in v:
                e["salu_insts"] = int(v["SQ_INSTS_SALU"])
            if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
                e["lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 3)
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                e["hbm_bytes"] = int(2 * 1024 * v["FETCH_SIZE"] + 1024 * v["WRITE_SIZE"])
            if v.get("TCP_TOTAL_CACHE_ACCESSES_sum") and "TCP_TCC_READ_REQ_sum" in v:
                e["l1_hit_rate_est"] = round(1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
            if "TCC_HIT_sum" in v and v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0) > 0:
                e["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
            if v.get("TD_TD_BUSY_sum") and v.get("SQ_BUSY_CU_CYCLES"):
                e["td_busy_frac"] = round(v["TD_TD_BUSY_sum"] / v["SQ_BUSY_CU_CYCLES"], 3)     # L1 data-return path busy
            if v.get("TA_TA_BUSY_sum") and v.get("SQ_BUSY_CU_CYCLES"):
                e["ta_busy_frac"] = round(v["TA_TA_BUSY_sum"] / v["SQ_BUSY_CU_CYCLES"], 3)     # L1 address path busy
            if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                e["l1_accesses"] = int(v["TCP_TOTAL_CACHE_ACCESSES_sum"])
            if k in d:
                e["rocprof_avg_ms"] = d[k]
            wdoc[k] = e
        doc["workloads"][wl] = wdoc
    path = os.path.join(ROOT, "profiles", "traffic.json")
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    print(open(path).read())


if __name__ == "__main__":
    main()
