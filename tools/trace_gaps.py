"""Reads a rocprofv3 kernel trace (…_kernel_trace.csv) and prints, per grid size of k_shadow (= per share size of
tools/share_cost.py), the mean duration of each kernel of a frame and the mean idle gap between consecutive kernels.
usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_" not in n:
        continue
    k = n[n.index("k_"):].split("<")[0].split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])))
rows.sort()
frames = []          # a frame = k_extend, k_shadow, k_replay_flat, k_resolve_u8 in a row
i = 0
while i + 3 < len(rows):
    ks = [rows[i + j][2] for j in range(4)]
    if ks == ["k_extend", "k_shadow", "k_replay_flat", "k_resolve_u8"]:
        frames.append(rows[i:i + 4]); i += 4
    else:
        i += 1
by = collections.defaultdict(list)
for fi, f in enumerate(frames):
    by[f[1][3]].append((fi, f))
for grid, fs in sorted(by.items(), reverse=True):
    fs = fs[5:]      # skip the warm-up frames of each share size
    if not fs:
        continue
    dur = [sum(f[j][1] - f[j][0] for _, f in fs) / len(fs) / 1e3 for j in range(4)]
    gap = [sum(f[j + 1][0] - f[j][1] for _, f in fs) / len(fs) / 1e3 for j in range(3)]
    nxt = [frames[fi + 1][0][0] - f[3][1] for fi, f in fs if fi + 1 < len(frames) and frames[fi + 1][1][3] == grid]
    span = [(frames[fi + 1][0][0] - f[0][0]) for fi, f in fs if fi + 1 < len(frames) and frames[fi + 1][1][3] == grid]
    print("k_shadow grid %9d: kernels %s us = %.1f;  gaps inside the frame %s us, to the next frame %.1f us;  frame period %.1f us"
          % (grid, " ".join("%.1f" % d for d in dur), sum(dur), " ".join("%.1f" % g for g in gap), sum(nxt) / max(len(nxt), 1) / 1e3, sum(span) / max(len(span), 1) / 1e3))
