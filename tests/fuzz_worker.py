"""Worker of tests/test_gpu_fuzz.py::test_fuzz_volume_in_worker_processes: every nproc-th case of the volume run, whole days without hooks.
    python tests/fuzz_worker.py <index> <nproc> <random cases> <medium cases>"""
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import test_gpu_fuzz as F      # noqa: E402

idx, nproc, n_random, n_medium = (int(x) for x in sys.argv[1:5])
done = 0
try:
    for k in range(idx, n_random, nproc):
        seed = 100_000 + k          # (seeds the parametrised tests do not use)
        F.run_case(seed, F.random_case(seed), whole_day=True)
        done += 1
    for k in range(idx, n_medium, nproc):
        seed = 5_000 + k
        F.run_case(seed, F.medium_case(seed), idle_cap=1024, whole_day=True)
        done += 1
except Exception:
    traceback.print_exc()
    print("FAILED at case", done, "of worker", idx)
    sys.exit(1)
print("WORKER DONE %d" % done)
