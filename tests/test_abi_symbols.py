"""CPU-side checks of the drop-in boundary: libvds.so builds/loads without a GPU, exports every
symbol include/vds.h declares, the ctypes struct matches the C struct, and the product fails loudly
(no CPU fallback) when no HIP device is visible."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="vds.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vds_[a-z_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    from vehicles_dispatch_simulator_amd import _lib
    _lib.build()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libvds.so does not export " + n
        assert n in _lib.SYMBOLS, "_lib.SYMBOLS lacks " + n
    assert set(_lib.SYMBOLS) == set(names)
    assert lib.vds_version() == (1 << 16)


def test_every_exported_symbol_is_declared_in_a_header():
    """`nm -D libvds.so`: every exported vds_* is declared in include/vds.h (the boundary) or include/vds_debug.h (test and
    instrumentation hooks: not the boundary), and the debug header names nothing that is not exported and bound as a test symbol."""
    import subprocess
    from vehicles_dispatch_simulator_amd import _lib
    _lib.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r" T (vds_[a-z_0-9]+)$", out, flags=re.M)))
    boundary, debug = declared_symbols(), declared_symbols("vds_debug.h")
    assert not set(boundary) & set(debug)
    assert exported == sorted(boundary + debug), (set(exported) ^ set(boundary + debug))
    assert set(debug) == set(_lib.TEST_SYMBOLS)
    # the product package calls no debug hook outside the explicit test switches of BatchedDispatchEnv (dense_debug=, close())
    pkg = os.path.join(ROOT, "vehicles_dispatch_simulator_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py") and f not in ("_lib.py", "env.py"):
            assert "vds_debug_" not in open(os.path.join(pkg, f)).read(), f


def test_config_struct_layout_matches_c():
    from vehicles_dispatch_simulator_amd import _lib
    lib = _lib.load()
    cfg = _lib.VdsConfig()
    lib.vds_config_init(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(_lib.VdsConfig)
    assert cfg.tick_minutes == 10 and cfg.pickup_reject_threshold == 600_000_000_000 and cfg.vehicles == 6000


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from vehicles_dispatch_simulator_amd import BatchedDispatchEnv
    import numpy as np
    with pytest.raises(Exception, match="no CPU fallback"):
        BatchedDispatchEnv(np.zeros((4, 4), np.int32), np.zeros(4, np.int32), np.zeros(2, np.int32), np.zeros(0, np.int32), replicas=1, vehicles=2)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "vehicles_dispatch_simulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "vds_oracle" not in txt, f
