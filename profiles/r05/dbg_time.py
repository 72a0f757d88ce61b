import os, sys, time
sys.path.insert(0, "tests")
import test_gpu_debug_builds as t
lib = os.path.join(t.ROOT, "build", "libvds_canary.so")
t0 = time.perf_counter()
out = t.run_worker(lib, "+canary", [(False, 0, 19, 150, None), (True, 0, 9, 40, None), (True, 3, 6, 40, None)])
print(out); print("total", time.perf_counter() - t0)
lib = os.path.join(t.ROOT, "build", "libvds_dbg.so")
t0 = time.perf_counter()
out = t.run_worker(lib, "+dbg", [(True, 0, 8, 40, None), (True, 0, 37, 25, None)])
print(out); print("total", time.perf_counter() - t0)
