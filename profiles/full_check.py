"""Full-scale parity: EVERY replica of the bench workload against the CPU oracle (per-order status / vehicle / wait
and counters, bit-exact).   python profiles/full_check.py [cfg2|cfg4] [replicas] [distinct days] [interleaved|blocked] [force_generic]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle.oracle import Oracle
from vehicles_dispatch_simulator_amd import workloads
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = workloads.didi_day("cfg2") if wl == "cfg2" else workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
init = w.vehicle_nodes(R)
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mapping = sys.argv[4] if len(sys.argv) > 4 else "interleaved"
days = workloads.distinct_days(w, D)
rd = (np.arange(R) % D) if mapping == "interleaved" else (np.arange(R) * D // R)
FG = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if D > 1:
    env = w.make_env(R, load=False, force_generic=FG)
    env.load_order_days(days, rd.astype(np.int32))
else:
    env = w.make_env(R, force_generic=FG)
env.reset(init)
# (three settling days: k_tick_dense's form per slot is chosen from the counters of a whole episode - vds_api.hip adapt_dense -, so the
# checked day is the one a bench times; VDS_DENSE_TICK_FORMS=alt alternates the two forms from the first day on)
import ctypes as C
for _ in range(3):
    env.run(env.T); env.sync(); env.reset_again()
env.run(env.T)
forms = np.zeros(env.T, dtype=np.uint8); nf = C.c_int32()
env._lib.vds_debug_tick_forms(env._h, forms.ctypes.data_as(C.c_void_p), env.T, C.byref(nf))
cn = env.counters()
oracles = [Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server, d[0], d[1], d[2], w.vehicles) for d in days]
t0 = time.time()
bad = 0
for r0 in range(0, R, 64):
    got = env.orders(r0, min(64, R - r0))
    for i in range(got["status"].shape[0]):
        r = r0 + i
        o = oracles[int(rd[r])]
        o.reset(init[r]); o.run_day()
        exp, oc = o.orders(), o.counters()
        n = exp["status"].size
        ok = all(np.array_equal(got[k][i][:n], exp[k]) for k in ("status", "vehicle", "wait")) and \
            (cn[r, 0], cn[r, 1], cn[r, 3], cn[r, 6], cn[r, 7]) == (oc["order_num"], oc["reject_num"], oc["wait_sum"], oc["sum_order_value"], oc["evals"])
        bad += 0 if ok else 1
print("%s (%d order day%s%s, %s): %d replicas checked against the oracle in %.0f s, mismatching replicas: %d" % (
    wl, D, "s" if D > 1 else "", ", %s map" % mapping if D > 1 else "", env.main_kernel() + ", %d of %d slots in the 16-lane form" % (int(forms[:nf.value].sum()), env.T), R, time.time() - t0, bad))
assert bad == 0
