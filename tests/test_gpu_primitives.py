"""GPU primitives: the DPP wave / row reductions and scans against numpy (bit-exact)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_dpp_reductions_match_numpy():
    from vehicles_dispatch_simulator_amd import _lib
    lib = _lib.load()
    cfg = _lib.VdsConfig()
    lib.vds_config_init(C.byref(cfg))
    h = C.c_void_p()
    assert lib.vds_create(C.byref(cfg), C.byref(h)) == 0, lib.vds_last_error(None)
    rng = np.random.default_rng(0)
    nw = 4096
    x = rng.integers(-2**31, 2**31 - 1, size=(nw, 64), dtype=np.int64).astype(np.int32)
    # adversarial rows: minimum planted at every lane, duplicates, INT_MAX / INT_MIN everywhere
    for l in range(64):
        x[l] = 1000
        x[l, l] = -5 - l
    x[64] = 2**31 - 1
    x[65] = -2**31
    out = np.zeros(nw, dtype=np.int32)
    rmin, rsum, rscan = (np.zeros((nw, 64), dtype=np.int32) for _ in range(3))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.vds_selftest_dpp(h, p(x), p(out), p(rmin), p(rsum), p(rscan), nw)
    assert rc == 0, lib.vds_last_error(h)
    np.testing.assert_array_equal(out, x.min(axis=1))
    rows = x.reshape(nw, 4, 16)
    np.testing.assert_array_equal(rmin.reshape(nw, 4, 16), np.broadcast_to(rows.min(axis=2, keepdims=True), rows.shape))
    lo = (rows & 0xFFFF).astype(np.int64)
    np.testing.assert_array_equal(rsum.reshape(nw, 4, 16), np.broadcast_to(lo.sum(axis=2, keepdims=True), rows.shape))
    np.testing.assert_array_equal(rscan.reshape(nw, 4, 16), np.cumsum(lo, axis=2))
    lib.vds_destroy(h)
