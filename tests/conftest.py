import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


# ---- worker processes started at the beginning of the session ------------------------------------------------------------------
# Two GPU tests deal hundreds of independent days to worker processes (the mode x fixture product of tests/test_gpu_parity.py, the fuzz
# volume of tests/test_gpu_fuzz.py): their cost is host work, and the pytest process itself keeps one core busy.  When such a test is
# among the selected ones its workers are started as soon as the collection is final and run BESIDE the other tests; the test collects
# them when its turn comes (``collect_workers(name)``), and starts them itself when it finds none (a single test run by hand).
_BACKGROUND = {}


def start_workers(name, argvs):
    """Start one process per argv (stdout + stderr into a temporary file each); idempotent per name."""
    import subprocess, tempfile
    if name in _BACKGROUND:
        return _BACKGROUND[name]
    procs = []
    for argv in argvs:
        f = tempfile.TemporaryFile(mode="w+")
        procs.append((subprocess.Popen(argv, stdout=f, stderr=subprocess.STDOUT, text=True), f))
    _BACKGROUND[name] = procs
    return procs


def collect_workers(name, timeout=1500):
    """[(returncode, output)] of the processes started under ``name``."""
    out = []
    for p, f in _BACKGROUND.pop(name):
        p.wait(timeout=timeout)
        f.seek(0)
        out.append((p.returncode, f.read()))
        f.close()
    return out


def pytest_collection_finish(session):
    if session.config.option.collectonly:
        return
    ids = [it.nodeid for it in session.items]
    try:
        if any(i.endswith("test_gpu_parity.py::test_mode_fixture_product_in_worker_processes") for i in ids):
            import test_gpu_parity
            start_workers("parity_product", test_gpu_parity.product_worker_argvs())
        if any(i.endswith("test_gpu_fuzz.py::test_fuzz_volume_in_worker_processes") for i in ids):
            import test_gpu_fuzz
            start_workers("fuzz_volume", test_gpu_fuzz.volume_worker_argvs())
    except Exception as e:      # (the tests start their workers themselves then)
        print("conftest: background workers not started: %r" % (e,))


def pytest_sessionfinish(session, exitstatus):
    for procs in _BACKGROUND.values():          # (a session that stopped early: the exact processes started above)
        for p, f in procs:
            if p.poll() is None:
                p.kill()
            f.close()
    _BACKGROUND.clear()
