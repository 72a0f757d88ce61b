"""One order stream per row (day mode 2 of k_tick_dense: more distinct days than the library can group): per slot, 8 lanes per replica
against 16 (each pinned with VDS_DENSE_LPR for a traced run).
  PYTHONPATH=$R VDS_DENSE_LPR=8|16 rocprofv3 --kernel-trace --output-format csv -d out_x -- python profiles/r06/tick_forms_days.py run 300
  python profiles/r06/tick_forms_days.py read out_8 out_16"""
import sys
sys.path.insert(0, ".")
if sys.argv[1] == "run":
    import numpy as np, torch
    from vehicles_dispatch_simulator_amd import workloads
    nd = int(sys.argv[2])
    w = workloads.didi_day("cfg2")
    R = 1024
    days = workloads.distinct_days(w, nd)
    env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream, load=False)
    env.load_order_days(days, (np.arange(R) % nd).astype(np.int32))
    env.set_run_groups(1)
    env.reset(w.vehicle_nodes(R))
    for _ in range(3):
        env.run(env.T); env.sync(); env.reset_again()
    print(env.main_kernel(), env.work())
else:
    import csv, glob
    def last_day(d):
        f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if "k_tick_dense" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        n = len(rows) // 3
        return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows][-n:], rows[-1]["Kernel_Name"][:64]
    (a, na), (b, nb) = last_day(sys.argv[2]), last_day(sys.argv[3])
    print(na, "%.2f ms per day |" % (sum(a) / 1000), nb, "%.2f ms per day" % (sum(b) / 1000))
    for t in range(0, len(a), 8):
        print("%3d" % t, " ".join("%3.0f/%3.0f" % (a[t + i], b[t + i]) for i in range(min(8, len(a) - t))))
    print("day with the better form per slot: %.2f ms" % (sum(min(x, y) for x, y in zip(a, b)) / 1000))
