"""Kernel trace of the hooked day graph (one chain; fixed actions, planes idle_pre + idle_now + supply + cl_orders): durations of the three
kernels of a slot and the gaps between them.   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python profiles/r05/hooked_trace.py ; then
python profiles/r05/hooked_trace.py --summarise <dir>"""
import sys, glob, csv
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_tick_dense", "k_pack_obs", "k_dispatch_dense"))]
    # the last hooked day: 148 x (tick, obs, dispatch)
    seq = rows[-3 * 148:]
    import collections
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(seq[:-1], seq[1:]):
        na = "tick" if "k_tick_dense" in a["Kernel_Name"] else ("obs" if "k_pack_obs" in a["Kernel_Name"] else "dispatch")
        nb = "tick" if "k_tick_dense" in b["Kernel_Name"] else ("obs" if "k_pack_obs" in b["Kernel_Name"] else "dispatch")
        dur[na].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
        gap[na + " -> " + nb].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    print("hooked day graph, one chain, under rocprofv3 --kernel-trace (the tracer adds to the gaps): mean us over the last day")
    for k, v in dur.items(): print("  kernel %-9s %6.1f us (min %.1f, max %.1f)" % (k, sum(v) / len(v), min(v), max(v)))
    for k, v in gap.items(): print("  gap %-20s %6.1f us" % (k, sum(v) / len(v)))
    tot = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3 / 148
    print("  slot (first start to last end / 148): %.1f us" % tot)
    sys.exit(0)
sys.path.insert(0, ".")
import numpy as np, torch
from vehicles_dispatch_simulator_amd import workloads
R, K = 1024, 8
w = workloads.didi_day("cfg2")
stream = torch.cuda.current_stream()
env = w.make_env(R, stream=stream.cuda_stream)
env.reset(w.vehicle_nodes(R))
T = env.T
n2c = np.asarray(w.city.node2cluster)
node_of = torch.tensor([int(np.flatnonzero(n2c == c)[0]) for c in range(env.C)], dtype=torch.int32, device="cuda")
ar = torch.arange(R, device="cuda", dtype=torch.int32)
acts = torch.full((R, K, 3), -1, dtype=torch.int32, device="cuda"); acts[:, :, 1] = 0; acts[:, :, 2] = node_of[0]
acts[:, 0, 0] = ar % env.C; acts[:, 0, 2] = node_of[((ar + 97) % env.C).long()]
acts[:, 1, 0] = (ar + 97) % env.C; acts[:, 1, 2] = node_of[(ar % env.C).long()]
import os
if os.environ.get("HOOK_NO_ACTIONS") == "1": acts[:, :, 0] = -1          # every action slot empty: what the launch costs by itself
env.set_run_groups(1, -1)
for _ in range(3):
    env.reset_again(); env.run_hooked(T, actions=acts, inflight=False)
torch.cuda.synchronize()
try: env.sync()
except Exception as e:
    if "skipped" not in str(e): raise
env.close()
