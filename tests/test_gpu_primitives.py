"""GPU primitives: the DPP wave-min reduction against numpy (bit-exact)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_wave_min_dpp_matches_numpy():
    from vehicles_dispatch_simulator_amd import _lib
    lib = _lib.load()
    cfg = _lib.VdsConfig()
    lib.vds_config_init(C.byref(cfg))
    h = C.c_void_p()
    assert lib.vds_create(C.byref(cfg), C.byref(h)) == 0, lib.vds_last_error(None)
    rng = np.random.default_rng(0)
    nw = 4096
    x = rng.integers(-2**31, 2**31 - 1, size=(nw, 64), dtype=np.int64).astype(np.int32)
    # adversarial rows: minimum planted at every lane, duplicates, INT_MAX everywhere
    for l in range(64):
        x[l] = 1000
        x[l, l] = -5 - l
    x[64] = 2**31 - 1
    x[65] = -2**31
    out = np.zeros(nw, dtype=np.int32)
    rc = lib.vds_selftest_wave_min(h, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), nw)
    assert rc == 0, lib.vds_last_error(h)
    np.testing.assert_array_equal(out, x.min(axis=1))
    lib.vds_destroy(h)
