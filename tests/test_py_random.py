"""vds_py_random_nodes (native MT19937) against CPython's stdlib `random`: the stream
`random.Random(seed).choice(range(N))` that InitVehiclesIntoCluster draws (simulator.py:253).  CPU only."""
import ctypes as C
import random

import numpy as np
import pytest

from vehicles_dispatch_simulator_amd import _lib, synth, workloads


@pytest.mark.parametrize("seed", [0, 1, 42, 1234, 2**31 - 1, 2**32 + 5, 2**63 + 11])
@pytest.mark.parametrize("N", [1, 2, 3, 4139, 4096, 65536, 100003])
def test_matches_stdlib_choice_stream(seed, N):
    lib = _lib.load()
    out = np.zeros(3000, dtype=np.int32)
    assert lib.vds_py_random_nodes(C.c_uint64(seed), N, out.size, None, out.ctypes.data_as(C.c_void_p)) == 0
    rng = random.Random(seed)
    exp = np.array([rng.choice(range(N)) for _ in range(out.size)], dtype=np.int32)
    np.testing.assert_array_equal(out, exp)


def test_retry_until_valid_consumes_the_same_draws():
    N, V = 500, 2000
    valid = (np.arange(N) % 3 != 0)
    got = workloads.native_vehicle_nodes(77, N, V, 3, valid)
    exp = synth.make_vehicle_nodes(77, N, V, 3, valid)
    np.testing.assert_array_equal(got, exp)
    assert valid[got].all()


def test_bad_arguments():
    lib = _lib.load()
    out = np.zeros(4, dtype=np.int32)
    none_valid = np.zeros(10, dtype=np.uint8)
    assert lib.vds_py_random_nodes(C.c_uint64(1), 10, 4, none_valid.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) != 0
    assert lib.vds_py_random_nodes(C.c_uint64(1), 0, 4, None, out.ctypes.data_as(C.c_void_p)) != 0
