"""ctypes binding of ``libvds.so`` (``include/vds.h``) - the only way the package reaches the GPU.

There is no CPU fallback: if the shared library is missing or no HIP device is visible the
calls raise, loudly."""
from __future__ import annotations

import ctypes as C
import os
import sys
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
def _lib_path():
    """``libvds.so`` next to this file; ``VDS_LIB=<name or path>`` selects another build (the instrumented ones live in
    ``<repo>/build/``: ``make -C csrc prof | dbg | canary``)."""
    name = os.environ.get("VDS_LIB", "libvds.so")
    if os.path.isabs(name):
        return name
    for d in (_HERE, os.path.join(os.path.dirname(_HERE), "build")):
        if os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return os.path.join(_HERE, name)


LIB_PATH = _lib_path()
CSRC = os.path.join(_HERE, "csrc")

NUM_COUNTERS = 8
COUNTER_NAMES = ("order_num", "reject_num", "matched", "wait_sum", "dispatch_num", "dispatch_cost", "sum_order_value", "evals")


class VdsConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("replicas", C.c_int32), ("vehicles", C.c_int32),
        ("tick_minutes", C.c_int32), ("neighbor_can_server", C.c_int32), ("pickup_reject_threshold", C.c_int64),
        ("idle_cap", C.c_int32), ("ring_cap", C.c_int32), ("ring_ticks", C.c_int32), ("far_cap", C.c_int32),
        ("force_generic", C.c_int32),
    ]


# every symbol include/vds.h declares: name -> (restype, argtypes)
_VP, _I32, _I64 = C.c_void_p, C.c_int32, C.c_int64
SYMBOLS = {
    "vds_version": (_I32, []),
    "vds_build_id": (C.c_char_p, []),
    "vds_config_init": (None, [C.POINTER(VdsConfig)]),
    "vds_create": (C.c_int, [C.POINTER(VdsConfig), C.POINTER(_VP)]),
    "vds_destroy": (C.c_int, [_VP]),
    "vds_last_error": (C.c_char_p, [_VP]),
    "vds_set_stream": (C.c_int, [_VP, _VP]),
    "vds_load_static": (C.c_int, [_VP, _VP, _I32, _VP, _I32, _VP, _VP, _I32]),
    "vds_load_orders": (C.c_int, [_VP, _VP, _VP, _VP, _I32]),
    "vds_load_order_days": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP]),
    "vds_load_orders_strided": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I64]),
    "vds_replica_ticks": (C.c_int, [_VP, _I32, C.POINTER(_I32), C.POINTER(_I32)]),
    "vds_set_idle_cap": (C.c_int, [_VP, _I32]),
    "vds_idle_cap": (C.c_int, [_VP]),
    "vds_num_ticks": (C.c_int, [_VP, C.POINTER(_I32)]),
    "vds_reset": (C.c_int, [_VP, _VP]),
    "vds_reset_again": (C.c_int, [_VP]),
    "vds_reset_random": (C.c_int, [_VP, _VP]),
    "vds_step": (C.c_int, [_VP]),
    "vds_apply_dispatch": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP]),
    "vds_apply_dispatch_ex": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vds_apply_dispatch_device": (C.c_int, [_VP, _I32, _VP]),
    "vds_advance": (C.c_int, [_VP]),
    "vds_run": (C.c_int, [_VP, _I32]),
    "vds_set_run_groups": (C.c_int, [_VP, _I32, _I32]),
    "vds_get_run_groups": (C.c_int, [_VP]),
    "vds_sync": (C.c_int, [_VP]),
    "vds_clock": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "vds_read_obs": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "vds_obs_device": (C.c_int, [_VP, C.POINTER(_VP)]),
    "vds_obs_device_planes": (C.c_int, [_VP, _I32, C.POINTER(_VP)]),
    "vds_obs_inplace": (C.c_int, [_VP, _I32, C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vds_run_hooked_invalidate": (C.c_int, [_VP]),
    "vds_supply_inplace": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_VP), C.POINTER(_I32)]),
    "vds_counters_device": (C.c_int, [_VP, C.POINTER(_VP)]),
    "vds_read_counters": (C.c_int, [_VP, _VP]),
    "vds_reduce_counters": (C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    "vds_reduce_counters_into": (C.c_int, [_VP, _VP]),
    "vds_profile_enable": (C.c_int, [_VP, _I32]),
    "vds_run_hooked": (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP]),
    "vds_set_replica_days": (C.c_int, [_VP, _VP]),
    "vds_profile_read": (C.c_int, [_VP, _VP, _I32, C.POINTER(_I32)]),
    "vds_read_orders": (C.c_int, [_VP, _I32, _I32, _VP, _VP, _VP]),
    "vds_read_lists": (C.c_int, [_VP, _I32] + [_VP] * 8),
    "vds_read_vehicles": (C.c_int, [_VP, _I32] + [_VP] * 5),
    "vds_read_work": (C.c_int, [_VP, _VP]),
    "vds_py_random_nodes": (C.c_int, [C.c_uint64, _I32, _I32, _VP, _VP]),
    "vds_cluster_cost_sums": (C.c_int, [_I32, _VP, _I32, _VP, _I32, _VP, _VP]),
    "vds_cluster_cost_sums_error": (C.c_char_p, []),
    "vds_main_kernel": (C.c_char_p, [_VP]),
    "vds_dfs_sequences": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, C.c_int64]),
}
TEST_SYMBOLS = {"vds_debug_read_span": (C.c_int, [_VP, _VP, _I32]), "vds_debug_read_err": (C.c_int, [_VP, _VP]), "vds_debug_graph_pool_size": (C.c_int, []), "vds_debug_tick_forms": (C.c_int, [_VP, _VP, _I32, _VP]), "vds_debug_check_guards": (C.c_int, [_VP]), "vds_debug_poke_guard": (C.c_int, [_VP]), "vds_debug_dense": (C.c_int, [_VP, _I32, _I32, _I32, _I32]), "vds_debug_ablate": (C.c_int, [_VP, _I32]), "vds_debug_read_prof": (C.c_int, [_VP, _VP]), "vds_selftest_dpp": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _I32])}

_lib = None


def build(force: bool = False) -> str:
    """Compile ``csrc/*.hip`` for gfx950 into ``libvds.so`` (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "vds.h"))
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "vds_debug.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", CSRC])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            try:                      # same image on the GPU box: hipcc is there, build in-tree on first use
                build(force=True)
            except Exception as e:
                raise RuntimeError("libvds.so is not built (%s) and building it failed (%s): run "
                                   "`make -C vehicles_dispatch_simulator_amd/csrc`; there is no CPU fallback" % (LIB_PATH, e))
        # One HIP/HSA runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64, and
        # a process that initialises the system runtime first (through libvds) and torch's copy second ends with
        # torch reporting "No HIP GPUs are available".  Importing torch first makes libvds bind to the runtime torch
        # loaded (same soname), which is the order bench.py and the RCCL path use anyway.
        if "torch" not in sys.modules and not os.environ.get("VDS_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in {**SYMBOLS, **TEST_SYMBOLS}.items():
            if os.environ.get("VDS_LIB") and not hasattr(lib, name):
                # an older build loaded on purpose for an A/B timing (profiles/ab.py): say so, never silently
                print("vehicles_dispatch_simulator_amd: %s (VDS_LIB) does not export %s" % (LIB_PATH, name), file=sys.stderr)
                continue
            fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
