"""k_tick_dense per slot in its two forms (32 rows x 8 lanes with 128-entry tables; 16 rows x 16 lanes with 256-entry tables), each pinned
for a traced run, next to the number of buckets that left the fast path in the first form (what adapt_dense decides from):
  cd /tmp; for f in a b; do [ $f = a ] && export VDS_DENSE_ADAPT=0 || { unset VDS_DENSE_ADAPT; export VDS_DENSE_LPR=16; }
    PYTHONPATH=$R rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tf_$f -- python $R/profiles/r06/tick_forms.py run cfg4; done
  python profiles/r06/tick_forms.py read gpurun_out/tf_a gpurun_out/tf_b cfg4"""
import sys
sys.path.insert(0, ".")
if sys.argv[1] == "run":
    import os, torch
    from vehicles_dispatch_simulator_amd import workloads
    wl = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
    w = workloads.didi_day("cfg2") if wl == "cfg2" else (workloads.didi_day("cfg4", neighbor=True, service_m=2000.0) if wl == "cfg4" else workloads.stress())
    R = 128 if wl == "cfg5" else 1024
    env = w.make_env(R, stream=torch.cuda.current_stream().cuda_stream)
    env.set_run_groups(1)
    env.reset(w.vehicle_nodes(R))
    prev, cnt = 0, []
    for t in range(env.T):          # slot by slot: the counter of buckets leaving the fast path
        env.step(); env.advance()
        s = env.work()["slow_path_buckets"]; cnt.append(s - prev); prev = s
    if os.environ.get("VDS_DENSE_ADAPT") == "0":
        open("/tmp/slow_%s.txt" % wl, "w").write(" ".join(map(str, cnt)))
    for _ in range(2):
        env.reset_again(); env.run(env.T); env.sync()
else:
    import csv, glob, os
    def last_day(d):
        f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if "k_tick_dense" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows][-148:], rows[-1]["Kernel_Name"][:60]
    (a, na), (b, nb) = last_day(sys.argv[2]), last_day(sys.argv[3])
    wl = sys.argv[4]
    cnt = [int(x) for x in open("/tmp/slow_%s.txt" % wl).read().split()]
    print(wl, "|", na, "%.2f ms per day |" % (sum(a) / 1000), nb, "%.2f ms per day" % (sum(b) / 1000))
    print("per slot: us in the first form / us in the second / buckets that left the fast path in the first")
    for t in range(0, 148, 6):
        print("%3d" % t, " ".join("%3.0f/%3.0f/%-5d" % (a[t + i], b[t + i], cnt[t + i]) for i in range(min(6, 148 - t))))
    print("day with the better form per slot: %.2f ms" % (sum(min(x, y) for x, y in zip(a, b)) / 1000))
    for lim in (16, 32, 48, 64, 96, 128, 192, 256, 384):
        print("  rule 'second form where more than %d buckets left': %.2f ms" % (lim, sum(y if c > lim else x for x, y, c in zip(a, b, cnt)) / 1000))
