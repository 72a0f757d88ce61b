"""Round 5 design study (CPU, numpy + the oracle as the checker): the neighbour-search tick as DEFERRED ACCEPTANCE.

The reference's MatchFunction (simulator.py:900-975) is a serial dictatorship: orders in id order, each takes its favourite
vehicle among those still idle - own cluster first (:924-933), the visit sequence of FindServerVehicleFunction (:978-996) only
when the own cluster is empty (:936).  With every vehicle ranking the orders the same way (by id), the serial-dictatorship
outcome is the unique stable matching, and order-proposing deferred acceptance reaches it whatever the order of proposals:
  * stamp[e] = lowest rank that has proposed to entry e so far (only ever decreases);
  * an order proposes to the first entry of its preference list with stamp > its rank; the previous holder (a later order) is bumped
    and proposes again: next entry of its own cluster alive for it, else - searching cluster - it turns dry.
The hybrid tick's first kernel leaves exactly such an intermediate state (own-cluster matches as if nothing were stolen).  This
script runs that state forward in ROUNDS (all active orders propose at once) on configs[3], checks every tick against the oracle
and prints what a parallel kernel would face: dry orders, rounds, bumps, rescans per tick.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from vehicles_dispatch_simulator_amd import synth, workloads  # noqa: E402

K = int(os.environ.get("WK_K", "3"))
w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
seq_off, seq = workloads.native_dfs_sequences(w.nbr_off, w.nbr_idx, w.depth_limit)
C, N, V = w.city.C, w.city.N, w.vehicles
cost = w.city.cost
n2c = w.city.node2cluster
capable = np.diff(w.nbr_off) > 0
replica = int(sys.argv[1]) if len(sys.argv) > 1 else 0
init = synth.init_vehicle_nodes(__import__("random").Random(w.veh_seed + replica), N, V)
o = Oracle(cost, n2c, w.nbr_off, w.nbr_idx, w.depth_limit, True, w.release_min, w.pickup, w.delivery, V)
o.reset(init)
T = o.num_ticks
rel = w.release_min
now0 = int(rel[0]) - 10
# cursor semantics: order i processed at tick = running max of (rel - now0) // 10 ; last order never
tick_of = np.maximum.accumulate(np.maximum((rel - now0) // 10, 0))
tick_of[-1] = 10 ** 9
stats = []
FREE = 1 << 30
t0 = time.time()
for t in range(T):
    L = o.lists(); veh = o.vehicles()
    now = now0 + 10 * t
    loc = veh["loc"].copy()
    lists = []
    for c in range(C):
        l = list(L["idle_veh"][L["idle_off"][c]:L["idle_off"][c + 1]])
        for j in range(L["arr_off"][c], L["arr_off"][c + 1]):
            if L["arr_min"][j] <= now:
                v = int(L["arr_veh"][j]); l.append(v); loc[v] = veh["dest"][v]
        lists.append(l)
    ids = np.flatnonzero(tick_of == t)              # ranks = position in ids
    n = ids.size
    pc = n2c[w.pickup[ids]]
    # ---- kernel 1: own-cluster matching as if nothing were stolen
    stamp = [np.full(len(l), FREE, dtype=np.int64) for l in lists]
    hold = {}                                       # rank -> (cluster, pos, cost)
    dry = []
    by_c = {}
    for rk in range(n):
        by_c.setdefault(int(pc[rk]), []).append(rk)
    before = {}                                     # (rank) -> orders of its bucket before it
    for c, rks in by_c.items():
        l = lists[c]
        locs = loc[l] if l else np.zeros(0, dtype=np.int64)
        for i, rk in enumerate(rks):
            before[rk] = i
            alive = np.flatnonzero(stamp[c] > rk) if len(l) else np.zeros(0, dtype=np.int64)
            if alive.size:
                cs = cost[w.pickup[ids[rk]], locs[alive]]
                j = alive[np.argmin(cs)]           # first strict minimum: lowest position among ties
                stamp[c][j] = rk; hold[rk] = (c, int(j), int(cs.min()))
            elif capable[c]:
                dry.append(rk)
    n_dry0 = len(dry)
    n_dry_buckets = len(set(int(pc[rk]) for rk in dry))
    # ---- deferred acceptance in rounds
    def pref_own(rk):
        c = int(pc[rk]); l = lists[c]
        alive = np.flatnonzero(stamp[c] > rk)
        if not alive.size:
            return None
        cs = cost[w.pickup[ids[rk]], loc[np.array(l)[alive]]]
        j = alive[np.argmin(cs)]
        return (c, int(j), int(cs.min()))

    def scan_dry(rk):
        """top-K of the visit sequence in the reference's order (cost, visit index, position) among entries with stamp > rk"""
        c0 = int(pc[rk]); best = []
        for vi, c in enumerate(seq[seq_off[c0] + 1:seq_off[c0 + 1]]):
            l = lists[c]
            if not l:
                continue
            alive = np.flatnonzero(stamp[c] > rk)
            if alive.size:
                cs = cost[w.pickup[ids[rk]], loc[np.array(l)[alive]]]
                best.extend(zip(cs.tolist(), [vi] * alive.size, alive.tolist(), [int(c)] * alive.size))
        best.sort()
        return best[:K], len(best)

    records = {}
    active = list(dry)
    is_dry = set(dry)
    rounds = bumps = rescans = scans = own_repicks = turned_dry = 0
    for rk in dry:
        records[rk] = list(scan_dry(rk)) + [0]     # [cands, total, cursor]
        scans += 1
    while active:
        rounds += 1
        prop = {}
        for rk in active:
            if rk in is_dry:
                rec = records[rk]
                tgt = None
                while True:
                    while rec[2] < len(rec[0]):
                        cs, vi, pos, c = rec[0][rec[2]]
                        if stamp[c][pos] > rk:
                            tgt = (c, pos, cs); break
                        rec[2] += 1
                    if tgt is not None or rec[1] <= len(rec[0]):
                        break
                    records[rk] = rec = list(scan_dry(rk)) + [0]; rescans += 1       # all kept candidates died: scan again
                    if not rec[0]:
                        break
            else:
                tgt = pref_own(rk); own_repicks += 1
                if tgt is None and capable[pc[rk]]:
                    is_dry.add(rk); turned_dry += 1
                    records[rk] = list(scan_dry(rk)) + [0]; scans += 1
                    rec = records[rk]
                    if rec[0]:
                        cs, vi, pos, c = rec[0][0]; tgt = (c, pos, cs)
            if tgt is None:
                hold.pop(rk, None)                  # rejected
                continue
            prop.setdefault((tgt[0], tgt[1]), []).append((rk, tgt))
        nxt = []
        for (c, pos), lst in prop.items():
            lst.sort()
            win, tgt = lst[0]
            old = stamp[c][pos]
            stamp[c][pos] = win; hold[win] = tgt
            if old != FREE:
                bumps += 1; hold.pop(int(old), None); nxt.append(int(old))
            nxt.extend(rk for rk, _ in lst[1:])    # lost the entry in this very round: propose again
        active = sorted(nxt)
    # ---- check against the oracle
    o.begin_tick()
    od = o.orders()
    for rk in range(n):
        i = ids[rk]
        if rk in hold:
            c, pos, cs = hold[rk]
            assert od["status"][i] == 1 and od["vehicle"][i] == lists[c][pos] and od["wait"][i] == cs, (t, rk)
        else:
            assert od["status"][i] == 2, (t, rk)
    o.end_tick()
    idle_tot = sum(len(l) for l in lists)
    stats.append((t, n, idle_tot, n_dry0, len(is_dry), rounds, bumps, own_repicks, turned_dry, scans, rescans, n_dry_buckets))
print("replica %d: %d ticks checked against the oracle in %.1f s, K = %d" % (replica, T, time.time() - t0, K))
a = np.array(stats)
names = "t orders idle dry0 dry rounds bumps own_repicks turned_dry scans rescans dry_buckets".split()
print("per tick  " + "  ".join("%s mean %.1f max %d" % (names[i], a[:, i].mean(), a[:, i].max()) for i in range(1, len(names))))
print("ticks with dry orders: %d; rounds histogram:" % int((a[:, 3] > 0).sum()), np.bincount(a[:, 5]).tolist())
heavy = a[np.argsort(-a[:, 4])[:10]]
print("ten heaviest ticks (t orders idle dry0 dry rounds bumps own_repicks turned_dry scans rescans):")
for r in heavy:
    print("  ", r.tolist())
