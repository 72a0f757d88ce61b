"""Worker of tests/test_gpu_parity.py::test_remaining_mode_fixture_pairs_in_worker_processes: the (fixture, mode, kind) items given as a JSON list,
each a full per-tick day (run_day: counters, observations, idle-list and arrival-dict order, per-order results).
    python tests/parity_worker.py '[["tiny_kmeans", "dense_tiny"], ...]'"""
import json
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import test_gpu_parity as P      # noqa: E402
from helpers import load_golden  # noqa: E402

done = 0
try:
    for name, mode, kind in json.loads(sys.argv[1]):
        g = load_golden(name)
        if kind == "ragged":         # R not a multiple of the workgroup's replica run; every replica its own vehicle seed
            P.run_day(g, R=37, same_init=False, list_every=29, **P.MODES[mode])
        else:
            P.run_day(g, R=3, same_init=bool(len(g["dispatch_log"])), **P.MODES[mode])
        done += 1
except Exception:
    traceback.print_exc()
    print("FAILED at pair", done)
    sys.exit(1)
print("WORKER DONE %d" % done)
