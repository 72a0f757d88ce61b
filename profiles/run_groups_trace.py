"""Summarise a rocprofv3 --kernel-trace csv of a grouped run (plain or neighbour-search tick): the GROUP launches of k_tick_dense /
k_tick_rows / k_dfs_walk (grid smaller than the kernel's largest grid in the trace = a launch over replicas / groups replicas), their mean
duration, and inside the spans they cover (a gap of more than 1 ms ends a span: days are replayed back to back) how much of the
time has 0 / 1 / 2+ of them in flight and the time per tick.

    python profiles/run_groups_trace.py <kernel_trace.csv> [ticks per day = 148]
"""
import csv, sys, collections
T = int(sys.argv[2]) if len(sys.argv) > 2 else 148
rows, queues = [], []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        # (the tick's first kernel - k_tick_dense on the dense layout, k_tick_rows on the wide one - is "rows" below)
        if "k_tick_rows" in n or "k_tick_dense" in n or "k_dfs_walk" in n:
            rows.append(("walk" if "k_dfs_walk" in n else "rows", int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))))
            queues.append(r.get("Queue_Id", "?"))
gmax = collections.defaultdict(int)
for k, s, e, g in rows: gmax[k] = max(gmax[k], g)
full = [x for x in rows if x[3] == gmax[x[0]]]
grp = sorted((x for x in rows if x[3] < gmax[x[0]]), key=lambda x: x[1])
for label, sel in (("full launches (all replicas)", full), ("group launches", grp)):
    d = collections.defaultdict(list)
    for k, s, e, g in sel: d[k].append(e - s)
    for k, v in d.items(): print("%s, %s: n %d mean %.1f us  min %.1f  max %.1f" % (label, k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
if not grp:
    print("no group launches in this trace"); sys.exit(0)
spans, cur = [], [grp[0]]
for x in grp[1:]:
    if x[1] - max(y[2] for y in cur[-8:]) > 1_000_000: spans.append(cur); cur = []
    cur.append(x)
spans.append(cur)
hist, wall, nl = collections.Counter(), 0, 0
for sp in spans:
    ev = sorted([(s, 1, k) for k, s, e, g in sp] + [(e, -1, k) for k, s, e, g in sp])
    n = {"rows": 0, "walk": 0}; tp = ev[0][0]
    for t, dlt, k in ev:
        hist[(min(n["rows"], 3), min(n["walk"], 4))] += t - tp
        tp = t; n[k] += dlt
    wall += ev[-1][0] - ev[0][0]; nl += len(sp)
print("%d spans, %.2f ms covered by %d group launches" % (len(spans), wall / 1e6, nl))
for key in sorted(hist): print("k_tick_dense / k_tick_rows in flight %d, k_dfs_walk in flight %d: %5.1f %%" % (key[0], key[1], 100.0 * hist[key] / max(1, wall)))
print("(a kernel tracer serialises the hardware queues far more than a free run does: how the chains overlap without one is in inflight.txt - "
      "launch spans stamped on the device by the instrumented build, profiles/r04/inflight.py)")

# gaps between consecutive group launches on the same hardware queue (a branch of the day graph runs on one queue): what a
# dependent kernel boundary costs inside a chain
byq = collections.defaultdict(list)
for x, q in zip(rows, queues):
    if x[3] < gmax[x[0]]: byq[q].append(x)
for q, v in sorted(byq.items()):
    v.sort(key=lambda x: x[1])
    gaps = collections.defaultdict(list)
    for a, b in zip(v, v[1:]):
        if b[1] - a[2] < 1_000_000: gaps[a[0] + "->" + b[0]].append(b[1] - a[2])
    busy = sum(x[2] - x[1] for x in v)
    print("queue %s: %d group launches, %.2f ms of kernel time; " % (q, len(v), busy / 1e6) + "; ".join(
        "%s gap median %.1f us, mean of the positive ones %.1f us, %d of %d negative (overlap on the queue), max %.1f" % (
            k, sorted(g)[len(g) // 2] / 1e3, sum(x for x in g if x > 0) / max(1, sum(1 for x in g if x > 0)) / 1e3, sum(1 for x in g if x < 0), len(g), max(g) / 1e3)
        for k, g in sorted(gaps.items())))
