#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec of the TJM hot path at L=64, chi=128 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

A *step* is one order-1 TJM time step (two-site TDVP sweep -> dissipation -> stochastic jump,
analog/analog_tjm.py:438-447 in the reference) of one batch of trajectories resident on the GPU.
A trajectory is 10 such steps plus one measurement of <Z_i> on every site (final-time sampling), so
trajectories/sec = trajectories_in_flight * K / 10 / elapsed.  Inputs (MPO, initial MPS, noise table,
uniforms) are resident in HBM / host memory before the timed region starts.

With --gpus N > 1 the script is launched by torch.distributed.run, one rank per GPU: trajectory indices
are sharded contiguously over ranks (weak scaling: every rank runs --batch trajectories), there is no
exchange during evolution and one RCCL all-reduce combines the observable sums at the end.
"""
from __future__ import annotations

import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")  # the CPU baseline leg mirrors the reference's 1-BLAS-thread workers

import argparse
import ctypes as C
import json
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEPS_PER_TRAJ = 10  # elapsed_time = 1.0 at dt = 0.1 (the headline configuration); main() rescales it for another --dt
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


WORKLOADS = {
    # name: (description, MPO builder, noise process name, gamma)   -- SURVEY section 8d
    "tfim": ("dissipative TFIM (J=1, g=0.5, D=3 MPO), pauli_z gamma=0.1 on every site", lambda api, L: api.MPO.ising(L, 1.0, 0.5), "pauli_z", 0.1),
    "xxz": ("XXZ chain (Jx=Jy=1, Jz=0.5, D=5 MPO), lowering gamma=0.05 on every site", lambda api, L: api.MPO.heisenberg(L, 1.0, 1.0, 0.5, 0.0),
            "lowering", 0.05),
    "lr-ising": ("long-range Ising (two-exponential fit of 1/r^3, D=4 MPO) with g=0.5, pauli_z gamma=0.05 on every site",
                 lambda api, L: api.MPO.long_range_ising(L, [0.8792, 0.1208], [0.0717, 0.5136], 0.5), "pauli_z", 0.05),
}


def build_inputs(L, chi, workload="tfim"):
    from yaqs_amd import api

    _, make_mpo, proc, gamma = WORKLOADS[workload]
    mpo = make_mpo(api, L)
    st = api.MPS(L, state="haar-random", pad=chi, rng=np.random.default_rng(1))
    st.normalize("B")
    noise = api.NoiseModel([{"name": proc, "sites": [i], "strength": gamma} for i in range(L)])
    return mpo, st, noise


def _cpu_step(args):
    """One order-1 TJM step of one trajectory on the CPU oracle; returns its own wall time."""
    L, chi, tol, traj, workload, tdvp_mode, dt = args
    from oracle import tjm_oracle as o
    from yaqs_amd import api  # host-side builders only (no GPU): the same MPO tensors as the GPU leg

    _, make_mpo, proc, gamma = WORKLOADS[workload]
    rng = np.random.default_rng(1)
    st = o.MPSState.haar(L, chi, rng)
    st.normalize("B")
    mpo = [np.asarray(w) for w in make_mpo(api, L).tensors]
    noise = [o.make_process(proc, [i], gamma) for i in range(L)]
    p = o.Params(dt=dt, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=tol, random_seed=42, tdvp_mode=tdvp_mode)
    t0 = time.perf_counter()
    o.tdvp(st, mpo, p)
    o.apply_dissipation(st, noise, dt, p)
    st = o.stochastic_process(st, noise, dt, p, o.trajectory_rng(42, traj))
    return time.perf_counter() - t0


def cpu_baseline(L, chi, tol, procs, workload="tfim", tdvp_mode="2site", dt=0.1):
    """The CPU oracle (a NumPy/SciPy port of the reference path), run the way the reference runs: `procs` forked
    single-BLAS-thread workers, one trajectory each (core/parallel_utils.py:331-390).  Bounded sample: ONE order-1 TJM
    step per worker, extrapolated to 10 steps per trajectory.  Must run before anything touches the GPU (fork)."""
    import multiprocessing as mp

    ncpu = len(os.sched_getaffinity(0))
    procs = max(1, min(procs, ncpu))
    t0 = time.perf_counter()
    if procs == 1:
        per = [_cpu_step((L, chi, tol, 0, workload, tdvp_mode, dt))]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            per = pool.map(_cpu_step, [(L, chi, tol, t, workload, tdvp_mode, dt) for t in range(procs)], chunksize=1)
    wall = time.perf_counter() - t0
    slowest = max(per)
    return {
        "value": procs / (STEPS_PER_TRAJ * slowest),
        "unit": "trajectories/sec",
        "cores": procs,
        "kind": "port",
        "sample": f"1 TJM step ({tdvp_mode} TDVP + dissipation + jump, workload {workload}) of 1 trajectory per worker at L={L}, chi={chi}, {procs} forked "
                  f"single-thread workers side by side: slowest {slowest:.1f} s, fastest {min(per):.1f} s (wall {wall:.1f} s incl. set-up), "
                  f"x{STEPS_PER_TRAJ} steps per trajectory; host has {ncpu} cores",
        "seconds_per_step": slowest,
        "per_core_value": 1.0 / (STEPS_PER_TRAJ * slowest),
    }


def pmc_traffic(L, chi, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    separate FETCH_SIZE and WRITE_SIZE runs of this script, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    PMC counters cannot be read from inside the timed run, so the number is only reported for the configuration it was
    collected on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except OSError:
        return None, "no PMC summary committed"
    if (rec.get("L"), rec.get("chi"), rec.get("batch")) != (L, chi, B):
        return None, "PMC summary was collected on a different configuration"
    return rec["traffic_bytes_per_launch"], rec["note"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories resident per GPU (the headline configuration has 1024 trajectories)")
    ap.add_argument("--length", type=int, default=64)
    ap.add_argument("--chi", type=int, default=128)
    ap.add_argument("--krylov-tol", type=float, default=1e-4)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="tfim", help="tfim is BASELINE.json's headline configuration")
    ap.add_argument("--tdvp-mode", choices=["2site", "1site"], default="2site")
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=32, help="forked single-thread CPU workers of the cpu_baseline leg (capped at the core count)")
    args = ap.parse_args()
    # stdout carries exactly one JSON line: everything else a library prints there (RCCL writes its version banner to
    # stdout when the communicator is created) is sent to stderr by pointing fd 1 at fd 2 for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    global STEPS_PER_TRAJ
    STEPS_PER_TRAJ = max(1, int(round(1.0 / args.dt)))  # a trajectory runs to elapsed_time = 1.0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_ref = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu_ref = cpu_baseline(args.length, args.chi, args.krylov_tol, args.cpu_procs, args.workload, args.tdvp_mode, args.dt)  # before torch / HIP

    import torch
    import torch.distributed as dist

    from yaqs_amd import _lib
    from yaqs_amd.api import is_pauli
    from yaqs_amd.engine import BatchEngine
    from yaqs_amd.tjm import trajectory_uniforms

    use_dist = args.gpus > 1 or world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    device = f"cuda:{local}"

    L, chi, B, K, W = args.length, args.chi, args.batch, args.steps, args.warmup
    mpo, st, noise = build_inputs(L, chi, args.workload)
    dt = args.dt
    eng = BatchEngine(L, chi, B, mpo.tensors, device=device)
    eng.set_params(dt=dt, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=args.krylov_tol, tdvp_mode=args.tdvp_mode)
    eng.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
    eng.load_state(st.tensors)
    traj = [rank * B + b for b in range(B)]
    u = np.stack([trajectory_uniforms(42, t, 2 * (K + W) + 4) for t in traj])
    pos = np.zeros(B, dtype=np.int64)
    lib = _lib.load()

    def step():
        eng.tdvp()
        eng.dissipate(dt)
        eng.set_uniforms(np.stack([u[np.arange(B), pos], u[np.arange(B), pos + 1]], axis=1))
        jumped, _ = eng.stochastic(dt)
        pos[:] += 1 + jumped

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        step()
    stats0 = eng.stats()
    lib.tjm_profile_cross_kernel(8)  # bracket every 8th launch of the dominant kernel with HIP events
    barrier()
    t0 = time.perf_counter()
    zsum = np.zeros(L)
    for k in range(K):
        step()
        if (k + 1) % STEPS_PER_TRAJ == 0 or k == K - 1:
            M = eng.site_moments()
            zsum += np.einsum("lb->l", (M[:, :, 0, 0] - M[:, :, 1, 1]).real)
    if world > 1:
        tz = torch.from_numpy(zsum).to(device)
        dist.all_reduce(tz, op=dist.ReduceOp.SUM)  # the only collective of the path (RCCL over xGMI)
        zsum = tz.cpu().numpy()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    stats1 = eng.stats()
    ms, nbytes, ns = C.c_double(0), C.c_double(0), C.c_int64(0)
    lib.tjm_profile_cross_kernel_read(C.byref(ms), C.byref(nbytes), C.byref(ns))
    lib.tjm_profile_cross_kernel(0)

    if rank == 0:
        total_traj = B * world
        value = total_traj * K / STEPS_PER_TRAJ / elapsed
        site_updates = total_traj * K * (2 * L - 3) / elapsed
        achieved = (nbytes.value / 1e9) / (ms.value / 1e3) if ms.value > 0 else None
        traffic, traffic_note = pmc_traffic(L, chi, B)
        out = {
            "metric": "trajectories/sec",
            "value": value,
            "unit": "trajectories/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",  # arithmetic type; the tensors are complex128 (interleaved re, im)
            "data": "synthetic",
            "config": {
                "workload": f"{L}-site {WORKLOADS[args.workload][0]}, chi={chi}, dt={dt:g}, "
                            f"order-1 TJM, {args.tdvp_mode} TDVP, svd_threshold=1e-12, krylov_tol={args.krylov_tol:g}, Haar chi-saturated initial MPS",
                "trajectories_in_flight_per_gpu": B,
                "steps_per_trajectory": STEPS_PER_TRAJ,
                "parallelism": f"trajectory-sharded x{world}",
                "storage": "complex128",
            },
            "site_updates_per_sec": site_updates,
            "counters_per_step": {k_: (stats1[k_] - stats0[k_]) / K for k_ in stats1},
            "mean_Z_site0": float(zsum[0] / total_traj),
            "roofline": {
                "bound": "hbm",
                "kernel": "jacobi_cross16x_kernel (X-rows block-pair step of the batched one-sided Jacobi SVD)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "avg_launch_us": (1e3 * ms.value / ns.value) if ns.value else None,
                "launches_sampled": int(ns.value),
            },
        }
        if cpu_ref is not None:
            out["cpu_baseline"] = cpu_ref
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
