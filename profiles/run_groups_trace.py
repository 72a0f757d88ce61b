"""Summarise a rocprofv3 --kernel-trace csv of a grouped run (plain or hybrid tick): per kernel name the mean duration, and how much of the
wall time has 0 / 1 / 2+ kernels in flight (do the groups' kernels really overlap?).

    python profiles/run_groups_trace.py <kernel_trace.csv>
"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "k_tick_rows" in n or "k_dfs_walk" in n:
            rows.append(("rows" if "k_tick_rows" in n else "walk", int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort(key=lambda x: x[1])
# keep the last day (148 ticks): the last quarter of the launches
rows = rows[len(rows) * 3 // 4:]
dur = collections.defaultdict(list)
for k, s, e, q in rows: dur[k].append(e - s)
for k, v in dur.items(): print("%s: n %d mean %.1f us  min %.1f  max %.1f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
print("queues:", collections.Counter(q for _, _, _, q in rows))
ev = []
for k, s, e, q in rows: ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
t_prev = ev[0][0]; n = {"rows": 0, "walk": 0}; hist = collections.Counter()
for t, d, k in ev:
    hist[(min(n["rows"], 2), min(n["walk"], 4))] += t - t_prev
    t_prev = t; n[k] += d
tot = sum(hist.values())
print("wall %.2f ms over %d launches" % (tot / 1e6, len(rows)))
for key in sorted(hist): print("rows in flight %d, walks in flight %d: %5.1f %%" % (key[0], key[1], 100.0 * hist[key] / tot))
