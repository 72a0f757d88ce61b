"""Per-tick durations of the hybrid tick's two kernels from a rocprofv3 --kernel-trace csv of a ONE-group run (VDS_RUN_GROUPS=1):
the last complete day's 148 launch pairs in order - which slots make up the day.

    python profiles/walk_ticks.py <kernel_trace.csv> [ticks per day = 148]
"""
import csv, sys
T = int(sys.argv[2]) if len(sys.argv) > 2 else 148
walk, rows = [], []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "k_dfs_walk" in n: walk.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        elif "k_tick_rows" in n: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
walk.sort(); rows.sort()
w = [d for _, d in walk[-T:]]; r = [d for _, d in rows[-T:]]
tot = sum(w) + sum(r)
print("last day: walk %.2f ms + rows %.2f ms = %.2f ms over %d slots" % (sum(w) / 1e6, sum(r) / 1e6, tot / 1e6, len(w)))
order = sorted(range(len(w)), key=lambda i: -w[i])
cum = 0
for rank, i in enumerate(order[:16]):
    cum += w[i] + r[i]
    print("slot %3d (minute %4d): walk %6.1f us  rows %5.1f us   cumulative share of the day %4.1f %%" % (i, i * 10, w[i] / 1e3, r[i] / 1e3, 100.0 * cum / tot))
for lo, hi in ((0, 100), (100, 200), (200, 400), (400, 10**9)):
    sel = [i for i in range(len(w)) if lo * 1000 <= w[i] < hi * 1000]
    print("walk %4d-%s us: %3d slots, %4.1f %% of the day" % (lo, "%4d" % hi if hi < 10**9 else " inf", len(sel), 100.0 * sum(w[i] + r[i] for i in sel) / tot))
