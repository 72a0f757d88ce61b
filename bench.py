#!/usr/bin/env python
"""Benchmark of the hot path on BASELINE.json's metric: env-steps x replicas / s.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A *step* is one pass of the tick path over one batch of synthetic input: one whole simulated
day (T = 148 ticks: Update -> Match -> SupplyExpect each) of R replicas per GPU on
BASELINE.json configs[1] (192 clusters, 10k vehicles, ~200k orders/day), inputs resident in
HBM.  Replicas shard across GPUs with no data-path collective (weak scaling: R per GPU is
fixed); the only collective is an RCCL all-reduce of the int64[8] aggregate counters per day.

The one JSON line printed by rank 0 also carries
  roofline      dominant kernel: HBM bytes per launch (rocprofv3 PMC figure of the SAME build, profiles/traffic.json; else
                the bytes the data layout has to move, and the line says which) / average launch duration (one HIP event
                pair on the kernel's stream around a whole day / launches) vs 8 TB/s HBM; next to it SURVEY.md 8(d)'s
                accounting bytes and the limiter the counters name;
  cpu_baseline  the CPU oracle (a C port of the reference algorithm, 1 thread) timed on this
                host on a bounded sample of the same workload;
  cpu_baseline_all_cores  the same oracle as one independent single-replica process per host core.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)


def cpu_baseline(w, init_rows, budget_s: float = 10.0, max_days: int = 2048):
    """The oracle (kind "port": C restatement of the reference's Python loop) on this host, one
    thread, whole days of single replicas of the same workload until ~budget_s of CPU work."""
    from oracle.oracle import Oracle
    o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server,
               w.release_min, w.pickup, w.delivery, w.vehicles)
    days, ticks, evals = 0, 0, 0
    t0 = time.perf_counter()
    while days < max_days and (time.perf_counter() - t0) < budget_s:
        o.reset(init_rows[days % len(init_rows)])
        ticks += o.run_day()
        evals += o.counters()["evals"]
        days += 1
    dt = time.perf_counter() - t0
    out = {"value": ticks / dt, "unit": "env-steps*replicas/s", "cores": 1, "kind": "port",
           "sample": "%d single-replica days (%d ticks, %.3g match evaluations) of the same workload in %.1f s" % (days, ticks, evals, dt),
           "match_evals_per_s": evals / dt}
    # "port" is NOT the reference's speed: the C restatement runs the same day ~1000x faster than the reference's pandas / Python
    # loop (calibrated in the build container on identical inputs, workloads.REFERENCE_CALIBRATION)
    from vehicles_dispatch_simulator_amd.workloads import REFERENCE_CALIBRATION
    cal = REFERENCE_CALIBRATION.get(w.name)
    if cal:
        out["reference_ratio"] = cal["reference_ratio"]
        out["reference_estimate"] = ticks / dt / cal["reference_ratio"]
        out["reference_ratio_note"] = ("the unmodified Python reference ran fixture %s in %.1f s where this oracle takes %.4f s (%s): the reference itself "
                                       "would do about value / reference_ratio env-steps/s on one core of this host" % (cal["fixture"], cal["reference_sim_s"], cal["oracle_s"], REFERENCE_CALIBRATION["where"]))
    return out


def cpu_baseline_all_cores(w, init_rows, budget_s: float = 5.0, max_procs: int = 256):
    """the same oracle as independent single-replica processes on every host core (the reference is
    single-threaded; multi-core = independent replica processes).  Fresh interpreters, no GPU state shared."""
    import shutil
    import subprocess
    import tempfile
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    procs_n = max(1, min(cores, max_procs))
    tmp = tempfile.mkdtemp(prefix="vds_cpu_")
    try:
        arrays = dict(cost=w.city.cost, node2cluster=w.city.node2cluster, nbr_off=w.nbr_off, nbr_idx=w.nbr_idx,
                      release_min=w.release_min, pickup=w.pickup, delivery=w.delivery, init=np.ascontiguousarray(init_rows),
                      scalars=np.array([w.depth_limit, int(w.neighbor_can_server), w.vehicles], dtype=np.int64))
        for k, v in arrays.items():
            np.save(os.path.join(tmp, k + ".npy"), np.ascontiguousarray(v))
        worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, worker, tmp, str(budget_s), str(i)], stdout=subprocess.PIPE, env=env)
                 for i in range(procs_n)]
        rate = evals_rate = 0.0
        days = ticks = 0
        for p in procs:
            out, _ = p.communicate(timeout=budget_s * 6 + 120)
            if p.returncode != 0:
                raise RuntimeError("cpu worker failed")
            d, t, e, dt = out.decode().split()
            days += int(d); ticks += int(t)
            rate += int(t) / float(dt); evals_rate += int(e) / float(dt)
        wall = time.perf_counter() - t0
        quota = ""
        try:                                     # a container CPU quota below the visible core count explains sub-linear scaling
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = "; cgroup cpu.max=%s" % ("unlimited" if q[0] == "max" else "%.1f cores" % (int(q[0]) / int(q[1])))
        except Exception:
            pass
        return {"value": rate, "unit": "env-steps*replicas/s", "cores": procs_n, "kind": "port",
                "sample": "%d processes x ~%.0f s, %d single-replica days (%d ticks) of the same workload; %.1f s wall incl. start-up%s" % (procs_n, budget_s, days, ticks, wall, quota),
                "match_evals_per_s": evals_rate}
    except Exception as e:       # a reported extra, never a reason to lose the bench line
        return {"value": None, "error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def leg_roofline(env_x, workload_key, replicas, stream, build_id):
    """The roofline object of a side leg (neighbour search, stress): one HIP event pair on the kernels' stream around a hook-less day in
    the leg's timed mode / launches = time per tick; HBM-side bytes per tick from the committed PMC passes of the SAME kernel build
    (profiles/traffic.json: read requests by size + WRITE_SIZE; null when the build differs); the VALU-issue floor from the same passes."""
    import torch
    T_x = env_x.T
    env_x.reset_again(); env_x.run(T_x); env_x.reset_again()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream); env_x.run(T_x); ev1.record(stream); ev1.synchronize()
    day_ms = ev0.elapsed_time(ev1)
    kern = env_x.main_kernel()
    from vehicles_dispatch_simulator_amd import workloads as _w
    side = _w.profile_side_data(ROOT, workload_key, replicas, kern)
    traffic, tb = side.get("hbm_bytes_per_launch"), side.get("traffic_build")
    same = traffic is not None and tb is not None and tb.split("+")[0] == build_id.split("+")[0]
    tick_s = day_ms * 1e-3 / T_x
    forms = env_x.tick_forms()
    out = {"bound": "hbm", "kernel": kern, "ticks": T_x, "day_kernel_ms": day_ms, "avg_launch_ms": tick_s * 1e3, "run_groups": env_x.run_groups(),
           "slots_in_16_lane_form": int(forms.sum()) if forms.size else (0 if "dense" in kern else None),
           "traffic": traffic if same else None, "traffic_build": tb, "traffic_build_matches": bool(same), "traffic_source": side.get("traffic_source"),
           "achieved": (traffic / tick_s / 1e9) if same else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": (traffic / tick_s / 1e9 / HBM_PEAK_GBS) if same else None}
    lim = side.get("limiter")
    if lim and (lim.get("build") or "").split("+")[0] == build_id.split("+")[0]:
        floor_s = lim["valu_insts_per_launch"] * lim["cycles_per_valu_inst"] / (lim["simds"] * lim["clock_hz"])
        out["limiter"] = {"valu_insts_per_launch": lim["valu_insts_per_launch"], "valu_floor_ms": floor_s * 1e3, "valu_frac": floor_s / tick_s,
                          "wave_wait_frac": lim.get("wave_wait_frac"), "source": lim.get("source")}
        if out["frac"] is not None and out["limiter"]["valu_frac"] > out["frac"]:
            out["bound"] = "valu-issue" if out["limiter"]["valu_frac"] > 0.6 else "latency (per-replica chain of dry orders: neither HBM nor VALU issue is busy)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed days (default 300: a 2.2 s timed region at configs[1])")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicas", type=int, default=1024, help="replicas PER GPU")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-neighbour-leg", action="store_true", help="skip the extra configs[3] (neighbour search) leg")
    ap.add_argument("--check", action="store_true", help="verify replica 0 against the oracle after the run")
    ap.add_argument("--distinct-days", type=int, default=16, help="extra leg: the same workload with this many different order days (0/1 = skip)")
    ap.add_argument("--no-distinct-all", dest="distinct_all", action="store_false", help="skip the leg with one order stream per replica")
    ap.add_argument("--distinct-all-days", type=int, default=128, help="distinct days of that leg (replica r replays day r %% N)")
    ap.add_argument("--no-hooked-leg", dest="hooked", action="store_false", help="skip the hooked-slot leg (step -> obs -> policy -> dispatch -> advance)")
    ap.add_argument("--no-fallbacks-leg", dest="fallbacks", action="store_false", help="skip the leg that times the documented fallback tick paths")
    ap.add_argument("--no-stress-leg", dest="stress", action="store_false", help="skip the configs[4] leg (2048 clusters / 100k vehicles / 2M orders, 128 replicas)")
    a = ap.parse_args()
    marks = [("start", time.perf_counter())]          # wall clock of the legs (leg_seconds in the line)
    def mark(name):
        marks.append((name, time.perf_counter()))

    import torch
    from vehicles_dispatch_simulator_amd import dist as vdist
    from vehicles_dispatch_simulator_amd import workloads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one rank per GPU (the driver's launch).  VDS_BENCH_BACKEND=gloo: the tests' way to run the SAME N > 1 path with two ranks on
    # a box with one GPU (RCCL refuses two ranks on one device): ranks then share devices round-robin
    backend = os.environ.get("VDS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible): one process per GPU" % (local_rank, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run the RCCL path is exercised even at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    R = a.replicas
    if a.workload == "cfg2":
        w = workloads.didi_day("cfg2")
        wname = "configs[1]: %d replicas/GPU x 192 k-means clusters, 4139 nodes, 10k vehicles, 200k synthetic orders/day, no neighbour search" % R
    elif a.workload == "cfg4":
        w = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
        wname = "configs[3]: %d replicas/GPU x 192 clusters with neighbour DFS depth 2, 10k vehicles, 200k orders/day" % R
    else:
        w = workloads.stress()
        wname = "configs[4] (stress): %d replicas/GPU x 2048 clusters, 16384 nodes, 100k vehicles, 2M synthetic orders/day" % R
    first, count = vdist.shard(R * world, world, rank)     # weak scaling: R replicas on every GPU, global replica ids
    assert count == R
    init = w.vehicle_nodes(R, first_replica=first)

    stream = torch.cuda.current_stream()
    env = w.make_env(R, device=local_rank, stream=stream.cuda_stream)
    env.reset(init)           # uploads start nodes once; they stay resident
    T = env.T
    totals = torch.zeros(8, dtype=torch.int64, device="cuda")

    def one_day():
        env.reset_again()
        env.run(T)
        env.reduce_counters_into(totals.data_ptr())
        if dist is not None:
            vdist.allreduce_counters(totals)   # RCCL over xGMI: aggregate reward/metrics only (64 B)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # settling (outside the W warm-up days, untimed): the library chooses k_tick_dense's form per slot from the per-slot counters of a
    # whole episode, which reach the host without synchronisation (vds_api.hip adapt_dense) - episode 1 is counted, the start of
    # episode 3 decides and rebuilds the day graph.  Synchronising between these days lets that happen here, not in the timed days.
    for _ in range(3):
        one_day()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        one_day()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_day()
    fence()
    elapsed = time.perf_counter() - t0
    env.sync()                # surfaces capacity errors, if any
    per_rank = None
    if dist is not None:
        # what the first real N > 1 run needs to explain itself: every rank's own wall time, its first GLOBAL replica and that
        # replica's start-node seed (+ a fingerprint of the nodes it drew), and the collective by itself
        import zlib
        mine_t = torch.tensor([elapsed / a.steps * 1e3, float(first), float(w.veh_seed + first), float(zlib.crc32(np.ascontiguousarray(init[0]).tobytes()))],
                              dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        allt = torch.stack(allt).cpu().numpy()
        ar_n = 20
        if backend == "nccl":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            probe = totals.clone()
            vdist.allreduce_counters(probe); torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(ar_n):
                vdist.allreduce_counters(probe)
            e1.record(stream); e1.synchronize()
            ar_us = e0.elapsed_time(e1) / ar_n * 1e3
            ar_how = "HIP events on the bench stream around %d back-to-back RCCL all-reduces of int64[8]" % ar_n
        else:
            probe = totals.cpu()
            vdist.allreduce_counters(probe)
            tq = time.perf_counter()
            for _ in range(ar_n):
                vdist.allreduce_counters(probe)
            ar_us = (time.perf_counter() - tq) / ar_n * 1e6
            ar_how = "host clock around %d back-to-back gloo all-reduces of int64[8] (host memory)" % ar_n
        vdist.ALLREDUCE_CALLS -= ar_n + 1      # (the probe is not part of the job's count)
        per_rank = {"ms_per_step": [float(x) for x in allt[:, 0]], "ms_per_step_min": float(allt[:, 0].min()), "ms_per_step_max": float(allt[:, 0].max()),
                    "first_replica": [int(x) for x in allt[:, 1]], "first_replica_seed": [int(x) for x in allt[:, 2]],
                    "first_replica_nodes_crc32": [int(x) for x in allt[:, 3]], "allreduce_us": ar_us, "allreduce_timing": ar_how}
        elapsed = vdist.max_over_ranks(elapsed, device="cuda")
    agg = totals.cpu().numpy()
    mark("import, workload, handle, warm-up, timed days")

    # ---- roofline pass (rank 0): per-launch duration of the dominant kernel, HIP events on its stream.  The day is
    #      stepped slot by slot so that the packed observations (device tensors) give the exact number of idle-list
    #      entries the kernel loaded: sum of PerMatchIdleVehicles over the (replica, cluster) buckets that had orders.
    roofline = None
    work = env.work()         # work of the last day on this rank
    build_id = (env._lib.vds_build_id() or b"").decode()
    if rank == 0:
        env.reset_again()
        env.profile(True)
        env.run(T)                                 # per-launch pass: HIP event pairs around every launch of the dominant kernel
        ms = env.profile_read(cap=T + 8)
        env.profile(False)
        # whole-day pass: ONE event pair on the kernel's stream around the T launches (the hipGraph vds_run replays), nothing
        # else on the stream in between; / launches = average launch duration incl. the kernel boundary, so that
        # avg_launch_ms x launches <= ms_per_step by construction (event pairs per launch add their own record overhead)
        env.reset_again()
        env.run(T); env.reset_again()              # (graph captured / warm)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        env.run(T)
        ev1.record(stream)
        ev1.synchronize()
        day_ms = ev0.elapsed_time(ev1)
        # the same day as ONE chain of launches over all replicas (VDS_RUN_GROUPS=1: what rocprofv3 --stats of profiles/<tag>/ times
        # kernel by kernel), same event-pair method, same process, same box: the line carries the two-chain gain itself
        groups_default = env.run_groups()
        one_chain_day_ms = day_ms
        if groups_default > 1:
            env.set_run_groups(1, 0)
            env.reset_again(); env.run(T); env.reset_again()
            ev0.record(stream); env.run(T); ev1.record(stream); ev1.synchronize()
            one_chain_day_ms = ev0.elapsed_time(ev1)
            env.set_run_groups(0, -1)              # library default again
            env.reset_again(); env.run(T)
        env.reset_again()                          # counting pass (untimed): the same day, observed slot by slot
        idle_loaded = torch.zeros((), dtype=torch.int64, device="cuda")
        busy_buckets = torch.zeros((), dtype=torch.int64, device="cuda")
        for _ in range(T):
            env.step()
            ob = env.obs_torch()                   # [5, R, C]: idle_pre, idle_now, supply, cl_orders, inflight
            has = ob[3] > 0
            idle_loaded += (ob[0] * has).sum()
            busy_buckets += has.sum()
            env.advance()
        if ms.size:
            kern = env.main_kernel()
            launches = int(ms.size)
            avg_s = day_ms * 1e-3 / launches
            alg_bytes = workloads.algorithmic_bytes(work, w.vehicles) / launches
            lay_bytes = workloads.layout_bytes(work, replicas=R, clusters=env.C, idle_loaded=int(idle_loaded.item()),
                                               busy_buckets=int(busy_buckets.item())) / launches
            side = workloads.profile_side_data(ROOT, a.workload, R, kern)
            traffic = side.get("hbm_bytes_per_launch")
            traffic_build = side.get("traffic_build")
            same_build = traffic is not None and traffic_build is not None and traffic_build.split("+")[0] == build_id.split("+")[0]      # (the kernel half of the id: host-only edits keep the evidence)
            # frac: HBM bytes per launch / average launch duration / 8 TB/s.  The bytes are the rocprofv3 PMC figure
            # (profiles/traffic.json: read requests by size + WRITE_SIZE, separate passes, calibrated against known byte counts in
            # profiles/ubench/bytes_calib.json) when it was measured on THIS build's kernels
            # (vds_build_id); otherwise the bytes the data layout has to move - a lower bound - and the line says so
            basis_bytes = traffic if same_build else lay_bytes
            roofline = {"bound": "hbm", "kernel": kern, "build": build_id,
                        "achieved": basis_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": basis_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                        "frac_basis": "pmc-traffic (profiles/traffic.json, same build)" if same_build else "layout-bytes (no PMC figure for this build)",
                        "traffic": traffic, "traffic_build": traffic_build, "traffic_build_matches": bool(same_build),
                        "traffic_basis": side.get("traffic_basis"), "traffic_source": side.get("traffic_source"),
                        "traffic_frac": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                        "layout_bytes_per_launch": lay_bytes, "layout_frac": lay_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                        # SURVEY 8(d)'s accounting unit (16 B per evaluation as if cost / location / idle entry came
                        # from HBM every time): NOT a lower bound for this layout, quoted for comparability only
                        "algorithmic_bytes_per_launch": alg_bytes,
                        "algorithmic_gbs": alg_bytes / avg_s / 1e9,
                        # avg_launch_ms: one TICK of the timed day = day_kernel_ms / launches (one HIP event pair around the day the
                        # timed loop replays).  vds_run issues a tick as `run_groups` concurrent launches of replicas / run_groups
                        # replicas each (parallel branches of the day graph) - bytes and time are per tick either way.
                        # event_pair_avg_launch_ms: ONE launch over all replicas by itself (the eager, profiled pass: an event pair
                        # per launch) - the figure rocprofv3 --stats of profiles/<tag>/kernel_stats.csv (VDS_RUN_GROUPS=1) shows
                        "avg_launch_ms": avg_s * 1e3, "launches": launches, "day_kernel_ms": day_ms,
                        "run_groups": groups_default, "slots_in_16_lane_form": int(env.tick_forms().sum()), "event_pair_avg_launch_ms": float(ms.mean()),
                        # one chain (one launch per tick over all replicas), one event pair around the day: the per-launch figure
                        # a kernel trace can confirm (profiles/<tag>/kernel_stats.csv: avg x launches <= this day)
                        "one_chain_day_kernel_ms": one_chain_day_ms, "one_chain_ms_per_tick": one_chain_day_ms / launches,
                        "frac_one_chain": basis_bytes / (one_chain_day_ms * 1e-3 / launches) / 1e9 / HBM_PEAK_GBS,
                        "two_chain_gain": one_chain_day_ms / day_ms,
                        "match_evals_per_s": work["evals"] / (day_ms * 1e-3)}
            lim = side.get("limiter")
            if lim:
                # what the counters say bounds the kernel: floor = VALU wave-instructions x cycles each
                # (profiles/ubench) / (SIMDs x clock); frac = floor / measured launch time
                floor_s = lim["valu_insts_per_launch"] * lim["cycles_per_valu_inst"] / (lim["simds"] * lim["clock_hz"])
                roofline["limiter"] = {"kind": "valu-issue + dependent-load latency (wavefronts wait %.0f %% of their cycles)" % (100 * (lim.get("wave_wait_frac") or 0)),
                                       "valu_floor_ms": floor_s * 1e3, "valu_frac": floor_s / avg_s,
                                       "valu_insts_per_launch": lim["valu_insts_per_launch"],
                                       "cycles_per_valu_inst": lim["cycles_per_valu_inst"],
                                       "wave_wait_frac": lim.get("wave_wait_frac"), "source": lim.get("source"),
                                       "build": lim.get("build"), "build_matches": (lim.get("build") or "").split("+")[0] == build_id.split("+")[0]}
                # `bound` names what the counters say limits the kernel: when the VALU floor is a larger share of the launch than the
                # HBM bytes are, the kernel is bound by instruction issue (index / compare / select work on wave64 VALUs, no MFMA);
                # achieved / peak / frac stay the HBM figures the contract asks for (how far below the memory roof the kernel runs)
                if roofline["limiter"]["valu_frac"] > roofline["frac"]:
                    roofline["bound"] = "valu-issue"
                    roofline["bound_note"] = ("VALU wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz) = %.0f %% of the launch (limiter.valu_frac) vs "
                                              "HBM bytes / 8 TB/s = %.0f %% (frac): instruction issue bounds this kernel, not HBM and not MFMA (no GEMM-shaped work); "
                                              "achieved / peak / frac are the HBM figures" % (100 * roofline["limiter"]["valu_frac"], 100 * roofline["frac"]))

    # ---- per-replica order days (rank 0, N = 1): the headline replays ONE day in every replica, which lets 16 replicas
    #      share staged order records and keeps per-order control flow wave-uniform.  The same workload with D distinct
    #      days (replica r replays day r % D; vds_load_order_days, k_tick_rows' per-row variant) is timed next to it.
    mark("roofline passes")
    per_days = None
    if rank == 0 and world == 1 and a.distinct_days > 1 and a.workload != "cfg5":
        days = workloads.distinct_days(w, a.distinct_days)
        per_days = {"distinct_days": a.distinct_days, "unit": "env-steps*replicas/s",
                    "note": "same city / vehicles / order count; interleaved: replica r replays day r %% %d (the library STORES the replicas "
                            "regrouped by day, every entry point maps the caller's replica index); blocked: day r // %d (the 16 replicas of a "
                            "workgroup on one day as given: shared-day code, day looked up per workgroup)" % (a.distinct_days, max(1, R // a.distinct_days))}
        for label, rmap in (("interleaved", np.arange(R) % a.distinct_days), ("blocked", np.minimum(np.arange(R) // max(1, R // a.distinct_days), a.distinct_days - 1))):
            env2 = w.make_env(R, device=local_rank, stream=stream.cuda_stream, load=False)
            env2.load_order_days(days, rmap.astype(np.int32))
            env2.reset(init)
            T2 = env2.T
            env2.reset_again(); env2.run(T2)
            torch.cuda.synchronize()
            nd = max(2, min(a.steps, 10))
            t1 = time.perf_counter()
            for _ in range(nd):
                env2.reset_again(); env2.run(T2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            env2.sync()
            per_days[label] = {"value": T2 * R * nd / dt2, "ms_per_step": dt2 / nd * 1e3, "steps": nd}
            env2.close()
        per_days["value"] = per_days["interleaved"]["value"]
        # one order stream PER REPLICA (day mode 2 of k_tick_dense: the ordinary RL case - every city its own episode): the replica ->
        # day map mixes days inside every group of 16 replicas and leaves fewer than 13 replicas per day, so the library can
        # neither keep nor regroup them into one-day workgroups.  128 distinct days by default (8 replicas per day; generating
        # and loading R = 1024 different days of 200k orders takes ~45 s of host time: --distinct-all-days 1024 does it)
        if a.distinct_all:
            nda = max(R // 12 + 1, min(a.distinct_all_days, R))
            daysR = workloads.distinct_days(w, nda)
            env2 = w.make_env(R, device=local_rank, stream=stream.cuda_stream, load=False)
            env2.load_order_days(daysR, (np.arange(R) % nda).astype(np.int32))
            env2.reset(init)
            T2 = env2.T
            env2.reset_again(); env2.run(T2)
            torch.cuda.synchronize()
            nd = max(2, min(a.steps, 10))
            t1 = time.perf_counter()
            for _ in range(nd):
                env2.reset_again(); env2.run(T2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            env2.sync()
            per_days["one_stream_per_replica"] = {"distinct_days": nda, "value": T2 * R * nd / dt2, "ms_per_step": dt2 / nd * 1e3, "steps": nd, "kernel": env2.main_kernel(),
                                           "slow_path_buckets_last_day": int(env2.work().get("slow_path_buckets", 0)),
                                           "vs_shared_day": (T2 * R * nd / dt2) / (T * R * world * a.steps / elapsed)}
            env2.close()

    # ---- neighbour-search mode (rank 0, N = 1, only next to the headline workload): BASELINE configs[3] - the same city with
    #      NeighborCanServer and a 2000 m service radius (DFS depth 2) - a few days, so that the line also shows the second tick path
    mark("per_replica_days")
    nbr = None
    if rank == 0 and world == 1 and a.workload == "cfg2" and not a.no_neighbour_leg:
        w4 = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
        env4 = w4.make_env(R, device=local_rank, stream=stream.cuda_stream)
        env4.reset(w4.vehicle_nodes(R))
        T4 = env4.T
        for _ in range(3):                # (settling: see the headline's)
            env4.reset_again(); env4.run(T4)
            torch.cuda.synchronize()
        nd4 = max(2, min(a.steps, 10))
        t1 = time.perf_counter()
        for _ in range(nd4):
            env4.reset_again(); env4.run(T4)
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t1
        env4.sync()
        nbr = {"workload": "configs[3]: %d replicas/GPU x 192 clusters with neighbour DFS depth 2, 10k vehicles, 200k orders/day" % R,
               "value": T4 * R * nd4 / dt4, "unit": "env-steps*replicas/s", "ms_per_step": dt4 / nd4 * 1e3, "steps": nd4,
               "kernel": env4.main_kernel(), "run_groups": env4.run_groups(),
               "slow_path_buckets_last_day": int(env4.work().get("slow_path_buckets", 0)),
               "roofline": leg_roofline(env4, "cfg4", R, stream, build_id),
               "evidence": "profiles/r06_cfg4 (rocprofv3 stats + PMC + per-tick trace of both kernels; bench.py --workload cfg4 --check: parity)"}
        env4.close()

    # ---- configs[4] (rank 0, N = 1, next to the headline): the stress city - 2048 clusters, 16384 nodes, 100k vehicles, 2M orders per replica
    #      and day - at 128 replicas on one GPU, three days
    mark("neighbour_search")
    stress = None
    if rank == 0 and world == 1 and a.workload == "cfg2" and a.stress:
        try:
            w5 = workloads.stress()
            R5 = 128
            env5 = w5.make_env(R5, device=local_rank, stream=stream.cuda_stream)
            env5.reset(w5.vehicle_nodes(R5))
            T5 = env5.T
            for _ in range(3):            # (the handle settles on its forms after the first episodes: DESIGN.md 4)
                env5.reset_again(); env5.run(T5)
                torch.cuda.synchronize()
            nd5 = 3
            t1 = time.perf_counter()
            for _ in range(nd5):
                env5.reset_again(); env5.run(T5)
            torch.cuda.synchronize()
            dt5 = time.perf_counter() - t1
            env5.sync()
            wk5 = env5.work()
            stress = {"workload": "configs[4] (stress): %d replicas/GPU x 2048 clusters, 16384 nodes, 100k vehicles, 2M synthetic orders/day" % R5,
                      "value": T5 * R5 * nd5 / dt5, "unit": "env-steps*replicas/s", "ms_per_step": dt5 / nd5 * 1e3, "steps": nd5, "replicas": R5,
                      "kernel": env5.main_kernel(), "run_groups": env5.run_groups(), "match_evals_per_s": wk5["evals"] / (dt5 / nd5),
                      "slow_path_buckets_last_day": int(wk5.get("slow_path_buckets", 0)),
                      "roofline": leg_roofline(env5, "cfg5", R5, stream, build_id),
                      "evidence": "profiles/r06_cfg5 (rocprofv3 stats + PMC); tests/test_gpu_stress_config.py: every replica against the oracle at R = 8"}
            env5.close()
        except Exception as e:       # (a reported extra)
            stress = {"error": repr(e)}

    # ---- the hooked slot (rank 0, N = 1): what an RL loop over R cities costs per slot - the reference's hook order
    #      (simulator.py:1057-1087: Match -> Reward / state hooks -> SupplyExpect -> DispatchFunction :893-898 -> next Update) with
    #      observations, policy and actions on the device: step -> obs_torch -> a small torch policy -> apply_dispatch_torch (K = 8
    #      moves per city and slot) -> advance.  Timed next to the same loop without the hook (step -> advance) and the hook-less
    #      vds_run above.
    mark("stress")
    hooked = None
    if rank == 0 and world == 1 and a.hooked and a.workload == "cfg2":
        K = 8
        n2c = np.asarray(w.city.node2cluster)
        some_node_of = torch.tensor([int(np.flatnonzero(n2c == c)[0]) if (n2c == c).any() else 0 for c in range(env.C)], dtype=torch.int32, device="cuda")
        zeros_k = torch.zeros((R, K), dtype=torch.int32, device="cuda")
        minus1 = torch.full((R, K), -1, dtype=torch.int32, device="cuda")

        def policy(ob):
            idle, supply, demand = ob[1], ob[2], ob[3]
            surplus = idle + supply - demand                             # [R, C] int32
            src = torch.topk(surplus, K, dim=1).indices                  # the K richest clusters of every city (distinct)
            dst = torch.topk(surplus, K, dim=1, largest=False).indices   # ... move one idle vehicle each towards the K poorest
            ok = (idle.gather(1, src) > 0) & (surplus.gather(1, src) - surplus.gather(1, dst) > 4)
            return torch.stack([torch.where(ok, src.int(), minus1), zeros_k, some_node_of[dst]], dim=2).contiguous()

        # three hooks: (a) a FIXED action tensor (what the boundary itself costs: k_pack_obs + k_dispatch_dense next to the tick),
        # (b) the torch policy captured once as a CUDA graph over the library's observation block (static address) and replayed
        # per slot (one submission), (c) the same policy issued eagerly (a dozen torch launches per slot: host-bound)
        obs_static = env.obs_torch()
        fixed_actions = torch.stack([minus1, zeros_k, some_node_of[:K].repeat(R, 1)], dim=2).contiguous()
        ar = torch.arange(R, device="cuda", dtype=torch.int32)
        fixed_actions[:, 0, 0] = ar % env.C                   # one move per city and slot (skipped where the list is empty: idle_pos 0) ...
        fixed_actions[:, 0, 2] = some_node_of[((ar + 97) % env.C).long()]      # ... to another cluster of that city,
        fixed_back = fixed_actions.clone()                    # and back on the odd slots (no cluster fills up over the day)
        fixed_back[:, 0, 0] = (ar + 97) % env.C
        fixed_back[:, 0, 2] = some_node_of[(ar % env.C).long()]
        actions_static = torch.zeros((R, K, 3), dtype=torch.int32, device="cuda")
        # the day-graph form (vds_run_hooked) applies ONE tensor every slot: both moves of the pair in it (steady state)
        fixed_both = fixed_actions.clone()
        fixed_both[:, 1, :] = fixed_back[:, 0, :]
        pol_graph, pol_graph_keep, graph_error = None, None, None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                for _ in range(2):
                    actions_static.copy_(policy(obs_static))
            stream.wait_stream(side)
            torch.cuda.synchronize()
            pol_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(pol_graph):
                actions_static.copy_(policy(obs_static))
            # the same capture with the graph kept (raw handle): embedded into the hooked day graph as a child graph
            pol_graph_keep = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(pol_graph_keep):
                actions_static.copy_(policy(obs_static))
        except Exception as e:       # (a reported extra)
            graph_error = repr(e)

        # the same with the observations read IN PLACE (vds_obs_inplace: idle_now / cl_orders are words of the bucket records, strided
        # views - no k_pack_obs per slot; a policy that does not need SupplyExpect): surplus = idle - demand
        # ... and SupplyExpect from the per-arrival-slot counters the tick kernels keep on request (vds_supply_inplace: ring[slot], both at
        # fixed addresses) - the SAME policy as above (surplus = idle + supply - demand), on a second handle: the planes cost the tick an
        # atomic per match, which the headline handle does not pay
        pol_inplace, inplace_error, env_s = None, None, None
        try:
            env_s = w.make_env(R, device=local_rank, stream=stream.cuda_stream)
            sup_ring, sup_slot = env_s.supply_inplace_torch()
            env_s.reset(init)
            for _ in range(3):            # (settling, hook-less days: see the headline's)
                env_s.run(env_s.T); torch.cuda.synchronize(); env_s.reset_again()
            views = env_s.obs_inplace_torch()

            def policy_inplace():
                idle, demand = views["idle_now"], views["cl_orders"]
                surplus = idle + sup_ring.index_select(0, sup_slot)[0] - demand
                src = torch.topk(surplus, K, dim=1).indices
                dst = torch.topk(surplus, K, dim=1, largest=False).indices
                ok = (idle.gather(1, src) > 0) & (surplus.gather(1, src) - surplus.gather(1, dst) > 4)
                return torch.stack([torch.where(ok, src.int(), minus1), zeros_k, some_node_of[dst]], dim=2).contiguous()

            side = torch.cuda.Stream()
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                for _ in range(2):
                    actions_static.copy_(policy_inplace())
            stream.wait_stream(side)
            torch.cuda.synchronize()
            pol_inplace = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(pol_inplace):
                actions_static.copy_(policy_inplace())
        except Exception as e:       # (a reported extra)
            inplace_error = repr(e)

        def hooked_day(kind):
            if kind.endswith("_inplace") or kind.endswith("_planes"):
                env_s.reset_again()
                if kind == "fixed_day_graph_inplace":         # tick -> k_dispatch_dense per slot: no observation pass (all four planes a policy reads are in place)
                    env_s.run_hooked(T, actions=fixed_both, idle_pre=False, idle_now=False, supply=False, cl_orders=False, inflight=False)
                elif kind == "policy_day_graph_inplace":
                    env_s.run_hooked(T, actions=actions_static, policy_graph=pol_inplace, idle_pre=False, idle_now=False, supply=False, cl_orders=False, inflight=False)
                else:                                         # packed planes as in fixed_day_graph, k_pack_obs taking `supply` from the counters
                    env_s.run_hooked(T, actions=fixed_both, inflight=False)
                return
            env.reset_again()
            if kind == "fixed_day_graph":         # one graph launch: T x (tick -> k_pack_obs -> k_dispatch_dense), replica groups as branches
                env.run_hooked(T, actions=fixed_both, inflight=False)
                return
            if kind == "policy_day_graph":        # ... with the torch policy as a child graph between observations and dispatch
                env.run_hooked(T, actions=actions_static, policy_graph=pol_graph_keep, inflight=False)
                return
            for ts in range(T):
                env.step()
                if kind == "fixed":
                    env.obs_torch(inflight=False)
                    env.apply_dispatch_torch(fixed_actions if ts % 2 == 0 else fixed_back)
                elif kind == "fixed_both":
                    env.obs_torch(inflight=False)
                    env.apply_dispatch_torch(fixed_both)
                elif kind == "graph":
                    env.obs_torch(inflight=False)           # k_pack_obs into the static block (the policy reads idle, supply, demand)
                    pol_graph.replay()
                    env.apply_dispatch_torch(actions_static)
                elif kind == "eager":
                    env.apply_dispatch_torch(policy(env.obs_torch(inflight=False)))
                env.advance()

        hook_errors = []

        def sync_skipping_refused_moves(e=None):
            # a move from an empty list is skipped and reported once (VDS_ESTATE, "... the action was skipped"): not an error of
            # this leg.  Anything else (capacity overflows: vehicles were lost) makes the timing meaningless: recorded, the leg says INVALID
            try:
                (e or env).sync()
            except Exception as e:
                if "the action was skipped" not in str(e):
                    hook_errors.append(str(e))

        res = {}
        kinds = ([("step_advance_only", "none"), ("engine_hook_fixed_actions", "fixed"), ("engine_hook_fixed_actions_both_moves", "fixed_both"), ("engine_hook_day_graph", "fixed_day_graph")]
                 + ([("torch_policy_graph", "graph")] if pol_graph is not None else []) + ([("torch_policy_day_graph", "policy_day_graph")] if pol_graph_keep is not None else [])
                 + ([("engine_hook_day_graph_obs_in_place", "fixed_day_graph_inplace"), ("engine_hook_day_graph_supply_planes", "fixed_day_graph_planes")] if env_s is not None else [])
                 + ([("torch_policy_day_graph_obs_in_place", "policy_day_graph_inplace")] if pol_inplace is not None else [])
                 + [("torch_policy_eager", "eager")])
        for label, kind in kinds:
            hooked_day(kind); torch.cuda.synchronize()            # warm
            sync_skipping_refused_moves(env_s if (kind.endswith("_inplace") or kind.endswith("_planes")) else env)
            nd = 2
            t1 = time.perf_counter()
            for _ in range(nd):
                hooked_day(kind)
            t_issue = time.perf_counter() - t1                    # host time to ISSUE the days (the stream runs behind)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            e_used = env_s if (kind.endswith("_inplace") or kind.endswith("_planes")) else env
            sync_skipping_refused_moves(e_used)
            res[label] = {"slot_us": dt / (nd * T) * 1e6, "host_issue_us_per_slot": t_issue / (nd * T) * 1e6, "value": T * R * nd / dt,
                          "dispatches_last_day": int(e_used.work().get("dispatches", 0))}
        hookless_us = elapsed / a.steps / T * 1e6
        best = "torch_policy_day_graph" if "torch_policy_day_graph" in res else ("torch_policy_graph" if "torch_policy_graph" in res else "torch_policy_eager")
        hooked = {"unit": "env-steps*replicas/s", "value": res[best]["value"], "K": K, "days": 2, "policy": best,
                  "slot_us": res[best]["slot_us"], "hookless_run_tick_us": hookless_us,
                  # the boundary without a policy: the hooked day as ONE graph (vds_run_hooked: tick -> observations -> dispatch per slot,
                  # replica groups as parallel branches) against the hook-less day graph of the headline
                  "engine_hook_vs_hookless_tick": res["engine_hook_day_graph"]["slot_us"] / hookless_us,
                  "engine_hook_call_by_call_vs_hookless_tick": res["engine_hook_fixed_actions"]["slot_us"] / hookless_us,
                  # observations read in place (vds_obs_inplace + vds_supply_inplace: idle_pre / idle_now / cl_orders / SupplyExpect with no
                  # k_pack_obs node; the supply counters cost the tick one atomic per match) and packed planes with `supply` taken from those counters
                  "engine_hook_obs_in_place_vs_hookless_tick": (res["engine_hook_day_graph_obs_in_place"]["slot_us"] / hookless_us) if "engine_hook_day_graph_obs_in_place" in res else None,
                  "engine_hook_supply_planes_vs_hookless_tick": (res["engine_hook_day_graph_supply_planes"]["slot_us"] / hookless_us) if "engine_hook_day_graph_supply_planes" in res else None,
                  "policy_slot_vs_hookless_tick": res[best]["slot_us"] / hookless_us,
                  "host_issue_us_per_slot": res[best]["host_issue_us_per_slot"],
                  "variants": res,
                  "note": "per slot: tick -> k_pack_obs -> policy -> k_dispatch_dense -> advance; wall time of 2 days incl. the per-day reset.  *_day_graph: "
                          "vds_run_hooked, one graph launch per day (the torch policy embedded as a child graph); the others: one call per step "
                          "(vds_step / vds_obs_device_planes / policy / vds_apply_dispatch_device / vds_advance); engine_hook_* = the boundary without a policy"}
        if graph_error:
            hooked["policy_graph_error"] = graph_error
        if inplace_error:
            hooked["obs_in_place_error"] = inplace_error
        if hook_errors:
            hooked["INVALID"] = "engine errors during the hooked days (timings are not those of a valid run): " + " | ".join(sorted(set(hook_errors)))
        if env_s is not None:
            env_s.close()

    # ---- the episode reset three ways (rank 0, N = 1): start nodes resident (vds_reset_again: what the timed steps use), drawn on
    #      the device per episode (vds_reset_random: one random.Random(seed) stream per replica, the reference's InitVehiclesIntoCluster
    #      draws bit for bit), generated on the host and uploaded (vds_reset: what an RL loop with fresh start nodes paid before)
    mark("hooked_slot")
    episode_reset = None
    if rank == 0 and world == 1 and a.hooked and a.workload in ("cfg2", "cfg5"):
        try:
            def tm(f, n):
                f(); env.sync()
                t1 = time.perf_counter()
                for _ in range(n):
                    f()
                env.sync()
                return (time.perf_counter() - t1) / n * 1e3
            seeds = (np.arange(R) + w.veh_seed + first).astype(np.uint64)
            t1 = time.perf_counter(); init2 = w.vehicle_nodes(R, first_replica=first); gen_ms = (time.perf_counter() - t1) * 1e3
            episode_reset = {"unit": "ms per episode reset of %d replicas" % R, "reset_again": tm(env.reset_again, 50), "reset_random_on_device": tm(lambda: env.reset_random(seeds), 10),
                             "reset_uploaded": tm(lambda: env.reset(init2), 3), "host_generation_native_mt19937": gen_ms}
            env.reset_random(seeds)
            episode_reset["same_nodes_as_host_generation"] = bool(all(np.array_equal(env.vehicles(r)["node"], init2[r]) for r in (0, R // 2, R - 1)))
        except Exception as e:       # (a reported extra)
            episode_reset = {"error": repr(e)}
        finally:
            env.reset(init)          # whatever happened above: the --check pass and the line below describe `init`

    # ---- the documented fallbacks (rank 0, N = 1): every tick path a configuration can end up on has a number next to the default's
    #      (DESIGN.md 5).  A few days each; the serial reference form of the neighbour search on 128 replicas (it takes seconds per day)
    mark("episode_reset")
    fallbacks = None
    if rank == 0 and world == 1 and a.fallbacks and a.workload == "cfg2":
        fallbacks = {"unit": "env-steps*replicas/s", "note": "same workload and replica count as the headline unless said; vds_config.force_generic selects the path"}
        def leg(wl, Rl, fg, nd, environ=None):
            saved = {k: os.environ.get(k) for k in (environ or {})}
            os.environ.update(environ or {})
            try:
                e = wl.make_env(Rl, device=local_rank, stream=stream.cuda_stream, force_generic=fg)
            finally:
                for k, v in saved.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            e.reset(wl.vehicle_nodes(Rl))
            Tl = e.T
            e.reset_again(); e.run(Tl); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nd):
                e.reset_again(); e.run(Tl)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            e.sync()
            out = {"value": Tl * Rl * nd / dt, "ms_per_step": dt / nd * 1e3, "steps": nd, "replicas": Rl, "kernel": e.main_kernel()}
            e.close()
            return out
        try:
            w4f = workloads.didi_day("cfg4", neighbor=True, service_m=2000.0)
            fallbacks["configs[1] wide layout, row-mapped kernel (force_generic 5: clusters above 255 nodes, V >= 2^24, costs beyond the dense keys)"] = leg(w, R, 5, 3)
            fallbacks["configs[1] generic kernel, one wavefront per bucket (force_generic 1: costs >= 2^23 or above the pickup window)"] = leg(w, R, 1, 2)
            fallbacks["configs[1] dense tick with the arrival ring instead of static arrival slots (VDS_DENSE_PULL=0)"] = leg(w, R, 0, 3, {"VDS_DENSE_PULL": "0"})
            fallbacks["configs[3] hybrid tick on the wide layout (VDS_DENSE_DFS=0: order days per replica, costs beyond a byte, orders without a static arrival slot)"] = leg(w4f, R, 0, 2, {"VDS_DENSE_DFS": "0"})
            fallbacks["configs[3] lower-bound rounds (force_generic 3: visit sequences over 256 clusters, costs >= 2^15)"] = leg(w4f, R, 3, 2)
            fallbacks["configs[3] serial reference form (force_generic 1), 128 replicas"] = leg(w4f, 128, 1, 1)
        except Exception as e:       # (a reported extra)
            fallbacks["error"] = repr(e)

    # ---- episode turnover (rank 0, N = 1): what an RL loop pays to change the order day between episodes (the reference: Reload,
    #      simulator.py:130-212).  (a) another replica -> day map over resident days (vds_set_replica_days: nothing rebuilt),
    #      (b) a new day on a loaded handle (vds_load_orders: host tables by counting sorts, the neighbour search's visit rows built
    #      on the device), (c) 16 new days (vds_load_order_days: the days built by worker threads)
    mark("fallbacks")
    episode_turnover = None
    if rank == 0 and world == 1 and a.hooked and a.workload in ("cfg2", "cfg4"):
        try:
            def wall(f, n):
                f(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n):
                    f()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / n * 1e3
            days16 = workloads.distinct_days(w, 16)
            env3 = w.make_env(R, device=local_rank, stream=stream.cuda_stream, load=False)
            maps = [np.random.default_rng(s).integers(0, 16, size=R).astype(np.int32) for s in range(4)]
            t1 = time.perf_counter(); env3.load_order_days(days16, (np.arange(R) % 16).astype(np.int32)); first16 = (time.perf_counter() - t1) * 1e3
            env3.reset(init)
            it = iter(range(10 ** 9))
            episode_turnover = {"unit": "ms per call, %d replicas" % R,
                                "load_order_days_16_first": first16,
                                "load_order_days_16_again": wall(lambda: env3.load_order_days(days16, (np.arange(R) % 16).astype(np.int32)), 3),
                                "set_replica_days_blocks_of_64": wall(lambda: env3.set_replica_days((np.arange(R) // 64 + next(it)) % 16), 10),
                                "set_replica_days_random_map": wall(lambda: env3.set_replica_days(maps[next(it) % 4]), 10)}
            env3.set_replica_days(maps[0]); env3.reset(init)
            episode_turnover["day_after_random_map_ms"] = wall(lambda: (env3.reset_again(), env3.run(env3.T)), 3)
            episode_turnover["kernel_after_random_map"] = env3.main_kernel()
            env3.close()
            episode_turnover["load_orders_on_a_loaded_handle"] = wall(lambda: env.load_orders(w.release_min, w.pickup, w.delivery), 5)
        except Exception as e:       # (a reported extra)
            episode_turnover = {"error": repr(e)}
        finally:
            env.reset(init)

    mark("episode_turnover")
    check = None
    if a.check and rank == 0:
        from oracle.oracle import Oracle
        env.reset_again()
        env.run(T)
        ncheck = min(2, R)
        got = env.orders(0, ncheck)
        ok = True
        for r in range(ncheck):
            o = Oracle(w.city.cost, w.city.node2cluster, w.nbr_off, w.nbr_idx, w.depth_limit, w.neighbor_can_server,
                       w.release_min, w.pickup, w.delivery, w.vehicles)
            o.reset(init[r]); o.run_day()
            exp = o.orders()
            ok = ok and all(np.array_equal(got[k][r], exp[k]) for k in ("status", "vehicle", "wait"))
        check = bool(ok)

    cpu = cpu_all = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(w, init[: min(R, 16)])
        cpu_all = cpu_baseline_all_cores(w, init[: min(R, 64)])

    mark("check, cpu_baseline")
    if rank == 0:
        env_steps = T * R * world * a.steps
        out = {
            # (first: survives a truncated record of the line)
            "aggregate_counters_last_day": {"order_num": int(agg[0]), "reject_num": int(agg[1]), "wait_sum": int(agg[2]), "evals": int(agg[4])},
            "metric": "env-steps x replicas / sec (192-cluster, 10k vehicles)" if a.workload != "cfg5" else "env-steps x replicas / sec (2048-cluster, 100k vehicles stress)",
            "value": env_steps / elapsed, "unit": "env-steps*replicas/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": wname, "replicas_per_gpu": R, "replicas_total": R * world, "ticks_per_step": T,
                       "step": "one simulated day (T ticks) of all replicas incl. episode reset and counter all-reduce",
                       "parallelism": "replica-sharded x%d, RCCL all-reduce of int64[8] metrics per day" % world},
            "match_evals_per_s": float(work["evals"]) * world * a.steps / elapsed if work["evals"] else None,
            "slow_path_buckets_last_day": int(work.get("slow_path_buckets", 0)),      # buckets the fast kernel handed to a slower path (rank 0)
            "roofline": roofline, "cpu_baseline": cpu, "build": build_id,
        }
        if dist is not None:
            out["collective"] = {"backend": dist.get_backend(), "world_size": world, "allreduce_calls": vdist.ALLREDUCE_CALLS,
                                 "payload": "int64[8] aggregate counters per day",
                                 "allreduce_us": per_rank["allreduce_us"], "allreduce_timing": per_rank["allreduce_timing"]}
            out["per_rank"] = {k: v for k, v in per_rank.items() if not k.startswith("allreduce")}
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
        if per_days is not None:
            out["per_replica_days"] = per_days
        if nbr is not None:
            out["neighbour_search"] = nbr
        if stress is not None:
            out["stress"] = stress
        if hooked is not None:
            out["hooked_slot"] = hooked
        if episode_reset is not None:
            out["episode_reset"] = episode_reset
        if episode_turnover is not None:
            out["episode_turnover"] = episode_turnover
        if fallbacks is not None:
            out["fallbacks"] = fallbacks
        if check is not None:
            out["parity_check_vs_oracle"] = check
        out["leg_seconds"] = {marks[i][0]: round(marks[i][1] - marks[i - 1][1], 2) for i in range(1, len(marks))}
        print(json.dumps(out))
    env.close()
    if dist is not None:
        dist.barrier()            # rank 0 still ran the roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
