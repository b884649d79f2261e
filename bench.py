#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec of the TJM hot path at L=64, chi=128 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

A *step* is one order-1 TJM time step (two-site TDVP sweep -> dissipation -> stochastic jump,
analog/analog_tjm.py:438-447 in the reference) of the trajectories resident on the GPU(s).
A trajectory is 10 such steps plus one measurement of <Z_i> on every site (final-time sampling), so
trajectories/sec = trajectories * K / 10 / elapsed.  Inputs (MPO, initial MPS, noise table,
uniforms) are resident in HBM / host memory before the timed region starts.

Multi-GPU: one process per GPU over RCCL.  Started under torch.distributed.run the script is one rank (RANK / LOCAL_RANK /
WORLD_SIZE from the environment); started directly with --gpus N > 1 it launches its N ranks itself (fresh child processes,
before anything touches the GPU) and relays rank 0's JSON line.  Trajectory indices are sharded contiguously over the ranks,
there is no exchange during evolution and one all-reduce combines the observable sums at the end.  Default for N > 1 is
STRONG scaling on BASELINE.json's ensemble (--trajectories 1024 in total, 1024 / N resident per GPU); --scaling weak keeps
--batch trajectories per GPU.
"""
from __future__ import annotations

import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")  # the CPU baseline leg mirrors the reference's 1-BLAS-thread workers

import argparse
import ctypes as C
import json
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEPS_PER_TRAJ = 10  # elapsed_time = 1.0 at dt = 0.1 (the headline configuration); main() rescales it for another --dt
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak = fp64 matrix (MFMA) peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


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


# ------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a NumPy/SciPy port of the reference path) on the GPU box's host cores
# ------------------------------------------------------------------------------------------------------------------------
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


_SLAB = {}  # inputs of the site-update sample, built once in the parent and inherited by the forked workers


def _slab_prepare(L, chi, tol, workload, dt):
    """Tensors and environments of ONE bulk site-update at the chain centre (SURVEY section 8d's unit of work)."""
    from oracle import tjm_oracle as o
    from yaqs_amd import api

    _, make_mpo, _, _ = WORKLOADS[workload]
    st = o.MPSState.haar(L, chi, np.random.default_rng(1))
    st.normalize("B")
    mpo = [np.asarray(w) for w in make_mpo(api, L).tensors]
    i = L // 2 - 1
    t = st.tensors
    # the sample only needs environments of the right shapes and an isometric gauge around the pair
    rb = o.right_environments(t, mpo)
    lb = o.identity_env(t[0].shape[1], mpo[0].shape[2])
    for k in range(i):
        lb = o.update_left_environment(t[k], t[k], mpo[k], lb)
    _SLAB.update(a=t[i], b=t[i + 1], w0=mpo[i], w1=mpo[i + 1], lb=lb, rb1=rb[i + 1],
                 p=o.Params(dt=dt, max_bond_dim=chi, svd_threshold=1e-12, krylov_tol=tol), dt=dt, tol=tol)


def _slab_run(n_updates):
    """n bulk site-updates (merge, two-site Krylov, truncated split, environment update, backward one-site Krylov) on the
    prepared tensors; returns seconds per site-update."""
    from oracle import tjm_oracle as o

    s = _SLAB
    t0 = time.perf_counter()
    for _ in range(n_updates):
        theta = o.merge_two_site(s["a"], s["b"])
        w2 = o.merge_mpo_tensors(s["w0"], s["w1"])
        theta = o.update_site(s["lb"], s["rb1"], w2, theta, 0.5 * s["dt"], s["tol"])
        a, b = o._split_tdvp(theta, s["p"], "right")
        lb1 = o.update_left_environment(a, a, s["w0"], s["lb"])
        o.update_site(lb1, s["rb1"], s["w1"], b, -0.5 * s["dt"], s["tol"])
    return (time.perf_counter() - t0) / n_updates


def cpu_allowance():
    """What the host lease allows this process: CPUs in the affinity mask, os.cpu_count(), the cgroup CPU quota (v2 cpu.max or v1
    cfs_quota_us / cfs_period_us) and the throttling counters of cpu.stat.  ``effective_cores`` = min(affinity, quota)."""
    info = {"affinity_cpus": len(os.sched_getaffinity(0)), "os_cpu_count": os.cpu_count(), "cgroup_quota_cores": None, "quota_source": None}

    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    v2 = read("/sys/fs/cgroup/cpu.max")
    if v2:
        parts = v2.split()
        info["quota_source"] = "/sys/fs/cgroup/cpu.max = " + v2
        if parts and parts[0] != "max" and len(parts) > 1 and float(parts[1]) > 0:
            info["cgroup_quota_cores"] = float(parts[0]) / float(parts[1])
    else:
        q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and per:
            info["quota_source"] = f"cpu.cfs_quota_us = {q}, cpu.cfs_period_us = {per}"
            if float(q) > 0 and float(per) > 0:
                info["cgroup_quota_cores"] = float(q) / float(per)
    eff = float(info["affinity_cpus"])
    if info["cgroup_quota_cores"]:
        eff = min(eff, info["cgroup_quota_cores"])
    info["effective_cores"] = eff
    return info


def cpu_throttle_counters():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            with open(path) as f:
                kv = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            return {k: int(v) for k, v in kv.items() if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time")}
        except (OSError, ValueError):
            continue
    return None


def cpu_baseline(L, chi, tol, procs_list, workload="tfim", tdvp_mode="2site", dt=0.1, full_steps=True):
    """The CPU oracle run the way the reference runs: P forked single-BLAS-thread workers, one trajectory each
    (core/parallel_utils.py:331-390).  Bounded sample.  Rows:
      * P = 1 and every P of ``procs_list`` that fits the time budget: ONE full order-1 TJM step per worker (slowest worker counts),
        extrapolated to 10 steps per trajectory;
      * every P including P = all cores: a SLAB sample - 3 bulk site-updates at the chain centre per worker, all workers side by
        side - from which the per-worker slow-down under contention is read and applied to the uncontended full-step time
        (a full step with every core busy takes minutes: measured once, 316-411 s at 256 workers).
    Must run before anything touches the GPU (fork)."""
    import multiprocessing as mp

    ncpu = len(os.sched_getaffinity(0))
    allowance = cpu_allowance()
    throttle0 = cpu_throttle_counters()
    ctx = mp.get_context("fork")
    rows = []
    t_all = time.perf_counter()
    # --- uncontended: one worker, the whole step
    if full_steps:
        t1 = _cpu_step((L, chi, tol, 0, workload, tdvp_mode, dt))
        rows.append({"cores": 1, "value": 1.0 / (STEPS_PER_TRAJ * t1), "seconds_per_step": t1, "how": "full step, one worker alone on the host"})
    else:
        # a full step of this configuration takes minutes on a core (chi = 256: 174 s measured in round 1): the bounded sample is the
        # slab - 3 bulk two-site updates at the chain centre - times the 2 (L - 1) updates of the two TDVP sweeps of a step.  The
        # dissipation and jump sweeps are NOT included: the CPU row is an UPPER bound of the CPU rate (the stricter direction).
        _slab_prepare(L, chi, tol, workload, dt)
        t1 = _slab_run(3) * 2 * (L - 1)
        rows.append({"cores": 1, "value": 1.0 / (STEPS_PER_TRAJ * t1), "seconds_per_step": t1,
                     "how": "full-step ESTIMATE, one worker alone: slab sample (3 bulk site-updates) x 2 (L - 1) updates per step, TDVP sweeps only (upper bound of the CPU rate)"})
    # --- full step with P workers side by side
    for P in (procs_list if full_steps else []):
        P = min(P, ncpu)
        if P <= 1 or any(r["cores"] == P and r["how"].startswith("full") for r in rows):
            continue
        with ctx.Pool(P) as pool:
            per = pool.map(_cpu_step, [(L, chi, tol, t, workload, tdvp_mode, dt) for t in range(P)], chunksize=1)
        rows.append({"cores": P, "value": P / (STEPS_PER_TRAJ * max(per)), "seconds_per_step": max(per), "fastest_worker_s": min(per),
                     "how": f"full step, {P} workers side by side (slowest worker)"})
    # --- slab sample: contention factor for every P up to all cores
    slab = []
    if tdvp_mode == "2site":
        _slab_prepare(L, chi, tol, workload, dt)
        base = _slab_run(3)
        ladder = sorted({p for p in (8, 16, 32, 64, 128, ncpu) if 1 < p <= ncpu})
        for P in ladder:
            with ctx.Pool(P) as pool:
                per = pool.map(_slab_run, [3] * P, chunksize=1)
            factor = max(per) / base
            slab.append({"cores": P, "seconds_per_site_update": max(per), "slowdown_vs_one_worker": factor})
            rows.append({"cores": P, "value": P / (STEPS_PER_TRAJ * t1 * factor), "seconds_per_step": t1 * factor,
                         "how": f"extrapolated: uncontended full step x slow-down of the 3-site-update slab sample at {P} workers"})
        _SLAB.clear()
    best = max(rows, key=lambda r: r["value"])
    measured_best = max((r for r in rows if r["how"].startswith("full")), key=lambda r: r["value"])  # ("full-step ESTIMATE" when full_steps is off)
    throttle1 = cpu_throttle_counters()
    allowance["cpu_stat_before"] = throttle0
    allowance["cpu_stat_after"] = throttle1
    if throttle0 and throttle1 and "nr_throttled" in throttle0:
        allowance["nr_throttled_during_cpu_leg"] = throttle1["nr_throttled"] - throttle0["nr_throttled"]
    # the reference's own default worker count: available_cpus() - 1 (core/parallel_utils.py:62-98, 216-220)
    ref_default_workers = max(1, ncpu - 1)
    ref_rows = [r for r in rows if r["cores"] >= min(ref_default_workers, ncpu) - 1 and r["cores"] <= ncpu]
    ref_default_row = min(ref_rows, key=lambda r: abs(r["cores"] - ref_default_workers)) if ref_rows else None
    quota = allowance.get("cgroup_quota_cores")
    best_label = (f"best row under the lease's CPU allowance (cgroup quota ~ {quota:.1f} cores of {ncpu} in the affinity mask)" if quota and quota < ncpu
                  else f"best row of the host ({ncpu} CPUs in the affinity mask, no cgroup quota below that)")
    return {
        "value": best["value"],
        "unit": "trajectories/sec",
        "cores": best["cores"],
        "kind": "port",
        "sample": f"oracle (NumPy/SciPy port of the reference path), forked single-BLAS-thread workers, one trajectory each, workload {workload}, "
                  f"L={L}, chi={chi}, {tdvp_mode} TDVP: 1 full TJM step per worker at P=1 and P in {list(procs_list)}, x{STEPS_PER_TRAJ} steps per "
                  f"trajectory; plus a 3-site-update slab sample at P up to all {ncpu} cores giving the contention slow-down; "
                  f"value = best whole-host row ({best['how']}); CPU leg took {time.perf_counter() - t_all:.0f} s",
        "host_cores": ncpu,
        "cpu_allowance": allowance,
        "best_row_is": best_label,
        "reference_default_workers": ref_default_workers,
        "reference_default_row": ({k: ref_default_row[k] for k in ("cores", "value", "seconds_per_step", "how")} if ref_default_row else None),
        "per_core_value": rows[0]["value"],  # one worker with the host to itself
        "best_measured_full_step": {k: measured_best[k] for k in ("cores", "value", "seconds_per_step")},
        "rows": rows,
        "slab_sample": {"seconds_per_site_update_one_worker": base, "by_workers": slab} if slab else None,
    }


def pmc_traffic(L, chi, B, kernel_tag):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes (profiles/r0*_pmc_traffic*.json: separate
    FETCH_SIZE and WRITE_SIZE runs of this script, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC counters
    cannot be read from inside the timed run: the number comes from a profile, is only reported for the configuration and the kernel
    (``kernel_tag``: "tjm32" = the complex64 instance, "tjm::" = the fp64 one) it was collected on, and the source says so."""
    for name in ("r06_pmc_traffic_zgemm4.json", "r05_pmc_traffic_zgemm4.json", "r05_pmc_traffic_quad64.json", "r04_pmc_traffic_c64q.json", "r04_pmc_traffic_c64.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
        except OSError:
            continue
        if (rec.get("L"), rec.get("chi")) != (L, chi) or (kernel_tag == "tjm32") != ("tjm32" in rec.get("kernel", "")) or \
                (kernel_tag == "zgemm4") != ("zgemm4" in rec.get("kernel", "")):
            continue
        # a launch covers the trajectories of ONE engine: scale the per-launch bytes of the collection run to that many
        scale = float(B) / float(rec.get("batch") or B)
        return rec["traffic_bytes_per_launch"] * scale, {"source": "committed profile", "file": "profiles/" + name, "steps_collected": rec.get("steps", 1),
                                                          "note": f"{rec['note']} (collected with {rec.get('batch')} trajectories per launch, scaled to "
                                                                  f"the {B} of this run's launches)"}
    return None, {"source": "none", "note": "no PMC summary committed for this configuration and kernel"}


def shard_rate(B):
    """Trajectories/s of ONE MI355X with B resident trajectories (profiles/r0*_shard_rates.json, measured with this script)."""
    for name in ("r06_shard_rates.json", "r05_shard_rates.json", "r04_shard_rates.json", "r03_shard_rates.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
        except OSError:
            continue
        row = rec.get("rates", {}).get(str(B))
        if row is not None:
            return {"trajectories_per_sec": row, "source": "profiles/" + name}
    return None


# ------------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run starts its own ranks
# ------------------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv, script=None, poll=0.2):
    """Start ranks 0..n-1 of ``script`` (default: this file) as fresh child processes with the torch.distributed.run environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relay rank 0's stdout, return the worst exit code.  Every child is polled: the
    first one that ends with a non-zero code takes the others down (SIGTERM, then SIGKILL after 10 s), so a rank that dies
    before the rendezvous cannot leave rank 0 waiting in the RCCL timeout."""
    import socket
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()  # a file, not a pipe: nobody has to drain it while the children are being polled
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            break
        time.sleep(poll)
    if failed is not None:
        print(f"[bench] rank {failed} ended with code {rcs[failed]}: stopping the other ranks", file=sys.stderr)
        live = [p for r, p in enumerate(procs) if rcs[r] is None]
        for p in live:
            p.terminate()
        deadline = time.time() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        rcs = [p.returncode for p in procs]
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    out0.close()
    if failed is not None:
        return abs(rcs[failed]) or 1
    return max(abs(rc) for rc in rcs)


# One flag per row of BASELINE.json's `configs` (1-based): the arithmetic BASELINE quotes the row in, the largest shard one MI355X holds.
# Config 2 is the headline (the defaults).  Config 5 is the circuit path (run_config5).
CONFIG_PRESETS = {
    3: dict(workload="xxz", chi=256, length=128, dt=0.05, dtype="complex64", trajectories=128, batch=128, steps=2, warmup=1),
    4: dict(workload="lr-ising", tdvp_mode="1site", chi=256, length=32, dt=0.05, trajectories=512, batch=512, steps=3, warmup=1, cpu_procs=[8]),
}


def run_config5(args):
    """BASELINE config 5: 64-site Trotter circuit (20 Ising layers = 1260 two-qubit gates), depolarising noise gamma = 0.001 after every
    gate on its sites, max_bond_dim 512, svd_threshold 1e-9, 8192 trajectories, complex64 (BASELINE quotes it in fp32) - through
    Simulator.run_circuit, the front end a user calls.  From |0...0> the bonds of this circuit stay below 8 (the cap of 512 is never
    approached): the line times the fused small-bond kernels.  --saturated: the regime the chi = 512 names - a Haar-saturated chi = 512
    input (268 MB per MPS in complex64), one wave of --trajectories (default 96), --layers Trotter layers (default 1): every gate is a 1024 x 1024 split
    (digital_tjm.py:455-533).  A 'step' is one Trotter layer."""
    import torch  # noqa: F401

    import yaqs_amd.tjm as tjm
    from yaqs_amd import _lib
    from yaqs_amd.api import DigitalSimParams, MPS, NoiseModel, Observable, Z, ising_trotter_layers

    L, sat = 64, bool(args.saturated)
    nlayers = args.layers if args.layers else (1 if sat else 20)
    ntraj = args.trajectories if args.trajectories != 1024 else (96 if sat else 8192)
    dtype = args.dtype if args.dtype_given else "complex64"
    layers = ising_trotter_layers(L, 1.0, 0.5, 0.1, nlayers)
    gates = sum(len(l.even) + len(l.odd) for l in layers)
    noise = NoiseModel([{"name": name, "sites": [i], "strength": 0.001} for i in range(L) for name in ("pauli_x", "pauli_y", "pauli_z")])
    p = DigitalSimParams(observables=[Observable(Z(), s) for s in range(L)], num_traj=ntraj, max_bond_dim=512, svd_threshold=1e-9, random_seed=42)
    if sat:
        st = MPS(L, state="haar-random", pad=512, rng=np.random.default_rng(1))
        st.normalize("B")
    else:
        st = MPS(L, state="zeros")
    lib = _lib.load(dtype)
    sim = tjm.Simulator(dtype=dtype, batch=(ntraj if sat else None))
    if not sat:  # warm-up: library load, first-touch allocations
        tjm.Simulator(dtype=dtype).run_circuit(MPS(L, state="zeros"), ising_trotter_layers(L, 1.0, 0.5, 0.1, 1),
                                               DigitalSimParams(observables=[Observable(Z(), 0)], num_traj=64, max_bond_dim=512, svd_threshold=1e-9, random_seed=1), noise)
    jw = np.zeros(4)
    lib.tjm_profile_cross_kernel(8 if sat else 0)
    lib.tjm_svd_work_read(jw.ctypes.data, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sim.run_circuit(st, layers, p, noise)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, nbytes, ns = C.c_double(0), C.c_double(0), C.c_int64(0)
    lib.tjm_profile_cross_kernel_read(C.byref(ms), C.byref(nbytes), C.byref(ns))
    lib.tjm_profile_cross_kernel(0)
    lib.tjm_svd_work_read(jw.ctypes.data, 0)
    max_bond = int(np.max(res.max_bond))
    f32 = dtype != "complex128"
    esz = 8 if f32 else 16
    peak = 2 * FP64_PEAK_TFLOPS if f32 else FP64_PEAK_TFLOPS
    if sat:
        # the Jacobi kernels of this regime (1024-column splits on the 8-column-block kernels of the large-bond path, 512-column centre
        # shifts on the tile kernel) are not all covered by the launch sampler: the figure is their executed flops over the WALL time of
        # the run - a lower bound of the kernels' own rate, which the rocprofv3 statistics of the same command resolve per kernel
        tfl = 28.0 * float(jw[0]) / 1e12 / el
        roof = {"bound": "fp32-valu" if f32 else "fp64-valu",
                "kernel": "Jacobi kernels of the large-bond path (svd_split_qr2: 1024 x 1024 gate splits; 1024 x 512 SVD centre shifts of the per-gate noise sweeps)",
                "achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": tfl / peak,
                "how": "28 real flops x rows x column pairs of every visited tile (device counter, all Jacobi kernels) / wall time of the whole run",
                "executed_jacobi_flops": 28.0 * float(jw[0]), "traffic": None}
    else:
        # small bonds: launch- / latency-bound fused kernels; the HBM figure is the algorithmic traffic of the gate updates (two site tensors
        # read and written per noisy gate at the largest bond met) over the wall time - far below any bandwidth bound by construction
        gbs = ntraj * gates * 2.0 * 2 * (2 * max_bond * max_bond * esz) / 1e9 / el
        roof = {"bound": "latency (launch-bound fused small-bond kernels: bonds <= %d)" % max_bond, "kernel": "small_sweep_kernel / svd_split_small_kernel",
                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                "how": "algorithmic bytes of the gate updates (two site tensors read + written per noisy gate, at the largest bond of the run) / wall time"}
    return {"metric": "trajectories/sec", "value": ntraj / el, "unit": "trajectories/sec", "n_gpus": 1, "steps": nlayers, "warmup": 0 if sat else 1,
            "ms_per_step": 1e3 * el / nlayers, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if f32 else "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 5{' (chi-saturated variant)' if sat else ''}: {L}-site Ising Trotter circuit, {nlayers} layers ({gates} two-qubit gates), "
                                   f"depolarising gamma=0.001 per gate and site, max_bond_dim=512, svd_threshold=1e-9, initial state "
                                   f"{'Haar chi=512 saturated' if sat else '|0...0>'}", "trajectories": ntraj, "storage": dtype, "largest_bond_met": max_bond,
                       "parallelism": "one MI355X"},
            "gate_updates_per_sec": ntraj * gates / el, "seconds": el, "roofline": roof}


def cpu_config5(saturated, nlayers):
    """Oracle digital_tjm on host cores, bounded: small bonds - 4 forked workers, one whole trajectory each; saturated - the first gates
    of one layer on one worker (a 1024 x 1024 zgesdd per gate), extrapolated per gate."""
    import multiprocessing as mp

    from oracle import tjm_oracle as o
    from yaqs_amd import api

    L = 64
    t_all = time.perf_counter()
    on = [o.make_process(name, [i], 0.001) for i in range(L) for name in ("pauli_x", "pauli_y", "pauli_z")]
    op = o.DigitalParams(observables=[o.Obs(o.PAULI["z"], s) for s in range(L)], max_bond_dim=512, svd_threshold=1e-9, random_seed=42)
    if not saturated:
        layers = [o.GateLayer(l.singles, l.even, l.odd, l.sample_points) for l in api.ising_trotter_layers(L, 1.0, 0.5, 0.1, nlayers)]
        global _C5
        _C5 = (on, op, layers)
        with mp.get_context("fork").Pool(4) as pool:
            per = pool.map(_c5_one, range(4), chunksize=1)
        return {"value": 4.0 / max(per), "unit": "trajectories/sec", "cores": 4, "kind": "port", "per_core_value": 1.0 / (sum(per) / 4),
                "sample": f"oracle digital_tjm, 4 forked single-BLAS-thread workers, one whole trajectory each ({nlayers} layers); slowest {max(per):.1f} s; "
                          f"CPU leg {time.perf_counter() - t_all:.0f} s", "cpu_allowance": cpu_allowance()}
    # One noisy gate of the oracle at the saturated chain centre takes 180.5 s on one core (measured once in the build container with
    # `TJM_CFG5_CPU_FULL=1 python bench.py --config 5 --saturated`: three gates, 571 s; every gate is a 1024 x 1024 zgesdd plus the
    # SVD / QR sweeps of its local noise over the chain) - too long for a bounded sample.  On the box the sample is ONE 1024 x 1024
    # zgesdd; the per-gate figure is scaled by its time against the 1.32 s the same call took where the 180.5 s were measured.
    import scipy.linalg

    full = api.ising_trotter_layers(L, 1.0, 0.5, 0.1, 1)[0]
    gates = (len(full.even) + len(full.odd)) * nlayers
    if os.environ.get("TJM_CFG5_CPU_FULL"):
        st = api.MPS(L, state="haar-random", pad=512, rng=np.random.default_rng(1))
        st.normalize("B")
        mid = [g for g in full.even if 28 <= g[0] <= 34][:3]
        lay = [o.GateLayer([], mid, [], full.sample_points)]
        s0 = o.MPSState([np.asarray(t, dtype=np.complex128) for t in st.tensors])
        t0 = time.perf_counter()
        o.digital_tjm(0, s0, on, op, lay)
        per_gate = (time.perf_counter() - t0) / max(1, len(mid))
        how = f"oracle digital_tjm on ONE worker: {len(mid)} noisy gates at the saturated chain centre, {per_gate:.1f} s per gate (measured here)"
    else:
        rng = np.random.default_rng(0)
        a = rng.standard_normal((1024, 1024)) + 1j * rng.standard_normal((1024, 1024))
        scipy.linalg.svd(a[:256, :256], lapack_driver="gesdd")  # (first call: library load)
        t0 = time.perf_counter()
        scipy.linalg.svd(a, full_matrices=False, lapack_driver="gesdd")
        t_svd = time.perf_counter() - t0
        per_gate = 180.5 * t_svd / 1.32
        how = (f"ESTIMATE: one 1024 x 1024 zgesdd on this host ({t_svd:.2f} s) x the oracle's measured cost of a noisy gate at the saturated chain centre in "
               f"units of that call (180.5 s per gate at 1.32 s per zgesdd, one core, build container; TJM_CFG5_CPU_FULL=1 measures it here: ~10 minutes)")
    return {"value": 1.0 / (per_gate * gates), "unit": "trajectories/sec", "cores": 1, "kind": "port", "per_core_value": 1.0 / (per_gate * gates),
            "sample": how + f", extrapolated to the {gates} gates of the run (edge gates are cheaper: an upper bound of the CPU time per trajectory); "
                            f"CPU leg {time.perf_counter() - t_all:.0f} s", "cpu_allowance": cpu_allowance()}


_C5 = None


def _c5_one(t):
    from oracle import tjm_oracle as o

    on, op, layers = _C5
    t0 = time.perf_counter()
    o.digital_tjm(t, o.MPSState.product(64, "zeros"), on, op, layers)
    return time.perf_counter() - t0


def shorten_prose(node):
    """Without --verbose the JSON line carries numbers: explanatory strings (how / note / sample ...) longer than 120 characters are cut
    to their first sentence (DESIGN.md section 5 has the full text).  Kernel and workload names stay."""
    if isinstance(node, dict):
        for k, v in list(node.items()):
            if isinstance(v, str) and len(v) > 120 and k not in ("workload", "kernel", "name"):
                cut = v.split(".  ")[0].split(". ")[0]
                node[k] = (cut[:117] + "...") if len(cut) > 120 else cut
            else:
                shorten_prose(v)
    elif isinstance(node, list):
        for v in node:
            shorten_prose(v)


def main():
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--config", type=int, default=2)
    cfg_no = pre.parse_known_args()[0].config
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="row of BASELINE.json's configs (1-based); 2 = the headline (default)")
    ap.add_argument("--saturated", action="store_true", help="config 5 only: Haar-saturated chi = 512 input, one wave, --layers Trotter layers")
    ap.add_argument("--layers", type=int, default=0, help="config 5: Trotter layers (default 20; 1 with --saturated)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trajectories", type=int, default=1024, help="ensemble of BASELINE.json's headline configuration (strong scaling splits it over the GPUs)")
    ap.add_argument("--batch", type=int, default=None, help="trajectories resident per GPU (default: --trajectories / N for strong scaling, --trajectories for weak)")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto", help="auto: strong for N > 1")
    ap.add_argument("--engines", type=int, default=4,
                    help="engines (host thread + HIP stream each) sharing one GPU's trajectories: their VALU-bound factorisations and "
                         "MFMA-bound contractions overlap on the device (measured on the MI355X: 1 -> 4 engines +6 %%)")
    ap.add_argument("--length", type=int, default=64)
    ap.add_argument("--chi", type=int, default=128)
    ap.add_argument("--krylov-tol", type=float, default=1e-4)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="tfim", help="tfim is BASELINE.json's headline configuration")
    ap.add_argument("--tdvp-mode", choices=["2site", "1site"], default="2site")
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--dtype", choices=["complex128", "complex64"], default="complex128",
                    help="complex128 is the reference's arithmetic and the headline; complex64 runs libtjm_hip_f32.so (fp32 arithmetic and storage)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI, one GPU per rank (the measurement).  gloo: the ranks share the GPUs that exist "
                         "(device = LOCAL_RANK mod device count) and reduce on the host - a functional check of the N > 1 path on a "
                         "box with fewer GPUs than ranks, not a scaling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true", help="keep the explanatory prose (how / note strings) in the JSON line; without it the line stays short")
    ap.add_argument("--cpu-procs", type=int, nargs="*", default=[32], help="worker counts of the full-step CPU rows besides P = 1")
    if cfg_no in CONFIG_PRESETS:
        ap.set_defaults(**CONFIG_PRESETS[cfg_no])
    args = ap.parse_args()
    args.dtype_given = any(a == "--dtype" or a.startswith("--dtype=") for a in sys.argv[1:])
    if args.config == 5:
        sys.stdout.flush()
        real_stdout5 = os.dup(1)
        os.dup2(2, 1)
        cpu5 = None if args.no_cpu_baseline else cpu_config5(args.saturated, args.layers if args.layers else (1 if args.saturated else 20))  # before torch / HIP
        out5 = run_config5(args)
        if cpu5 is not None:
            out5["cpu_baseline"] = cpu5
            out5["speedup_vs_cpu"] = {"vs_cpu_row": out5["value"] / cpu5["value"], "vs_one_core": out5["value"] / cpu5["per_core_value"]}
        os.write(real_stdout5, (json.dumps(out5) + "\n").encode())
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # nothing has touched the GPU in this process

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
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running on {world} rank(s)", file=sys.stderr)
    cpu_ref = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_ref = cpu_baseline(args.length, args.chi, args.krylov_tol, args.cpu_procs, args.workload, args.tdvp_mode, args.dt,
                               full_steps=not (args.chi > 128 and args.tdvp_mode == "2site"))  # before torch / HIP

    import torch
    import torch.distributed as dist

    from yaqs_amd import _lib
    from yaqs_amd.api import is_pauli
    from yaqs_amd.engine import BatchEngine
    from yaqs_amd.tjm import trajectory_uniforms

    gloo = args.dist_backend == "gloo"
    if gloo:
        local = local % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        world = dist.get_world_size()  # n_gpus below is what RCCL saw
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    red_device = "cpu" if gloo else device  # where the collectives run

    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    if args.batch is not None:
        B = args.batch
        if scaling == "strong":
            B = args.batch // world
    else:
        B = args.trajectories // world if scaling == "strong" else args.trajectories
    if B < 1:
        raise SystemExit("fewer than one trajectory per GPU")
    E = max(1, min(args.engines, B))
    L, chi, K, W = args.length, args.chi, args.steps, args.warmup
    mpo, st, noise = build_inputs(L, chi, args.workload)
    dt = args.dt
    sizes = [B // E + (1 if k < B % E else 0) for k in range(E)]
    first = rank * B
    engines, trajs = [], []
    for k, nb in enumerate(sizes):
        eng = BatchEngine(L, chi, nb, mpo.tensors, device=device, stream=torch.cuda.Stream(device=device) if E > 1 else None, dtype=args.dtype)
        eng.set_params(dt=dt, svd_threshold=1e-12, max_bond_dim=chi, krylov_tol=args.krylov_tol, tdvp_mode=args.tdvp_mode)
        eng.set_noise(noise.processes, [is_pauli(q) for q in noise.processes])
        eng.load_state(st.tensors)
        engines.append(eng)
        trajs.append(list(range(first, first + nb)))
        first += nb
    lib = _lib.load(args.dtype)  # the library the engines run on (its Jacobi launch sampler)

    class Drive:
        """One host thread per engine: the steps of its trajectories, the measurement at the end of every trajectory."""

        def __init__(self, eng, traj):
            self.eng, self.nb = eng, eng.B
            self.u = np.stack([trajectory_uniforms(42, t, 2 * (K + W) + 4) for t in traj])
            self.pos = np.zeros(self.nb, dtype=np.int64)
            self.zsum = np.zeros(L)
            self.err = None
            self.step_s = []  # wall seconds of every timed step of this engine
            self.warm_s = []  # ... and of the warm-up steps before them

        def step(self):
            e, ar = self.eng, np.arange(self.nb)
            e.tdvp()
            e.dissipate(dt)
            e.set_uniforms(np.stack([self.u[ar, self.pos], self.u[ar, self.pos + 1]], axis=1))
            jumped, _ = e.stochastic(dt)
            self.pos[:] += 1 + jumped

        def run(self, n, measure):
            try:
                for k in range(n):
                    t_step = time.perf_counter()
                    self.step()
                    # stochastic() has read the jump decisions back: the step's work on this stream is done
                    (self.step_s if measure else self.warm_s).append(time.perf_counter() - t_step)
                    if measure and ((k + 1) % STEPS_PER_TRAJ == 0 or k == n - 1):
                        M = self.eng.site_moments()
                        self.zsum += np.einsum("lb->l", (M[:, :, 0, 0] - M[:, :, 1, 1]).real)
            except BaseException as exc:  # surfaced by run_all
                self.err = exc

    drives = [Drive(e, t) for e, t in zip(engines, trajs)]

    def run_all(n, measure):
        if len(drives) == 1:
            drives[0].run(n, measure)
        else:
            th = [threading.Thread(target=d.run, args=(n, measure)) for d in drives]
            for t in th:
                t.start()
            for t in th:
                t.join()
        for d in drives:
            if d.err is not None:
                raise d.err

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_all(W, False)
    stats0 = [e.stats() for e in engines]
    for e in engines:
        e.profile(True)
    lib.tjm_profile_cross_kernel(8)  # bracket every 8th launch of the Jacobi tile kernels with HIP events on their engine's stream
    lib.tjm_profile_gemm(8)          # ... and of the fp64 GEMM kernel (zgemm4_kernel; executed tiles counted on the device)
    lib.tjm_profile_qr_apply(8)      # ... and of the block-reflector apply of the QR preconditioner
    jw = np.zeros(4)
    lib.tjm_svd_work_read(jw.ctypes.data, 1)  # reset the executed-work counters of the tiled Jacobi kernels
    mx = np.zeros(10)
    lib.tjm_svd_mixed_read(mx.ctypes.data, 1)  # ... and of the mixed-precision two-site split
    barrier()
    t0 = time.perf_counter()
    run_all(K, True)
    zsum = sum(d.zsum for d in drives)
    if world > 1:
        tz = torch.from_numpy(zsum).to(red_device)
        dist.all_reduce(tz, op=dist.ReduceOp.SUM)  # the only collective of the path (RCCL over xGMI)
        zsum = tz.cpu().numpy()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    stats1 = [e.stats() for e in engines]
    prof = [e.profile_read() for e in engines]
    ms, nbytes, ns = C.c_double(0), C.c_double(0), C.c_int64(0)
    lib.tjm_profile_cross_kernel_read(C.byref(ms), C.byref(nbytes), C.byref(ns))
    ms32, nb32, ns32 = C.c_double(0), C.c_double(0), C.c_int64(0)
    lib.tjm_profile_cross_kernel_read_c64(C.byref(ms32), C.byref(nb32), C.byref(ns32))  # the complex64 instance (mixed-precision split)
    lib.tjm_profile_cross_kernel(0)
    gp = np.zeros(6)
    lib.tjm_profile_gemm_read(gp.ctypes.data)
    lib.tjm_profile_gemm(0)
    qr_own, qr_c64 = np.zeros(5), np.zeros(5)
    lib.tjm_profile_qr_apply_read(qr_own.ctypes.data, qr_c64.ctypes.data)
    lib.tjm_profile_qr_apply(0)
    lib.tjm_svd_work_read(jw.ctypes.data, 0)
    lib.tjm_svd_mixed_read(mx.ctypes.data, 0)

    # ---- ONE engine alone on the device for one more step, EVERY launch of the dominant kernel bracketed by HIP events: the launch
    # duration the roofline fraction is formed from is a measurement of the kernel by itself, not a time slice of four overlapping
    # streams times an attribution factor.  The other engines idle; their states are not used again.
    iso = None
    if True:
        jw_i, mx_i = np.zeros(4), np.zeros(10)
        lib.tjm_svd_work_read(jw_i.ctypes.data, 1)
        lib.tjm_svd_mixed_read(mx_i.ctypes.data, 1)
        lib.tjm_profile_cross_kernel(1)
        lib.tjm_profile_gemm(1)
        lib.tjm_profile_qr_apply(1)
        torch.cuda.synchronize()
        t_i = time.perf_counter()
        drives[0].run(1, False)
        torch.cuda.synchronize()
        t_i = time.perf_counter() - t_i
        if drives[0].err is not None:
            raise drives[0].err
        drives[0].warm_s.pop()  # (run() logged the isolated step as a warm-up step: the per-step lists below hold the W + K steps of the run only)
        ms_i, nb_i, ns_i = C.c_double(0), C.c_double(0), C.c_int64(0)
        lib.tjm_profile_cross_kernel_read(C.byref(ms_i), C.byref(nb_i), C.byref(ns_i))
        ms32_i, nb32_i, ns32_i = C.c_double(0), C.c_double(0), C.c_int64(0)
        lib.tjm_profile_cross_kernel_read_c64(C.byref(ms32_i), C.byref(nb32_i), C.byref(ns32_i))
        lib.tjm_profile_cross_kernel(0)
        gp_i = np.zeros(6)
        lib.tjm_profile_gemm_read(gp_i.ctypes.data)
        lib.tjm_profile_gemm(0)
        qr_own_i, qr_c64_i = np.zeros(5), np.zeros(5)
        lib.tjm_profile_qr_apply_read(qr_own_i.ctypes.data, qr_c64_i.ctypes.data)
        lib.tjm_profile_qr_apply(0)
        lib.tjm_svd_work_read(jw_i.ctypes.data, 0)
        lib.tjm_svd_mixed_read(mx_i.ctypes.data, 0)
        iso = {"step_s": t_i, "trajectories": drives[0].nb,
               # one 64 x 64 output tile x one unit of K = 3 real matrix-core products x 2 flops x 64 x 64 (three-product complex multiplication)
               "gemm": {"ms": float(gp_i[0]), "samples": int(gp_i[1]), "launches": int(gp_i[2]), "bytes": float(gp_i[4]), "flops": 6.0 * 64 * 64 * float(gp_i[3])},
               "f64": {"ms": ms_i.value, "samples": int(ns_i.value), "bytes": nb_i.value, "flops": 28.0 * float(jw_i[0])},
               "c64": {"ms": ms32_i.value, "samples": int(ns32_i.value), "bytes": nb32_i.value, "flops": 28.0 * float(mx_i[6])},
               # block-reflector apply of the QR preconditioner (complex64 in both libraries: the fp64 library's instance serves the mixed split)
               "qr": {"ms": float((qr_c64_i if args.dtype == "complex128" else qr_own_i)[0]), "flops": float((qr_c64_i if args.dtype == "complex128" else qr_own_i)[1]),
                      "samples": int((qr_c64_i if args.dtype == "complex128" else qr_own_i)[2])}}

    if rank == 0:
        total_traj = B * world
        value = total_traj * K / STEPS_PER_TRAJ / elapsed
        site_updates = total_traj * K * (2 * L - 3) / elapsed
        f32 = args.dtype != "complex128"
        peak64, peak32 = FP64_PEAK_TFLOPS, 2 * FP64_PEAK_TFLOPS  # vector = matrix rate for both types on this part: 78.6 / 157.3 TFLOP/s
        peak = peak32 if f32 else peak64
        valu_bound, mfma_bound = ("fp32-valu", "mfma-f32") if f32 else ("fp64-valu", "mfma-f64")
        # ---- algorithmic work of this rank's timed region (SURVEY section 8d formulas; cMAC = 8 real flops) -----------------
        d_, D = 2, max(int(w.shape[3]) for w in mpo.tensors)
        n = d_ * chi
        F_mv2 = 8.0 * (2 * d_ ** 2 * D * chi ** 3 + d_ ** 4 * D ** 2 * chi ** 2)
        F_mv1 = 8.0 * (2 * d_ * D * chi ** 3 + d_ ** 2 * D ** 2 * chi ** 2)
        F_svd = 88.0 * n ** 3   # nominal zgesdd count of the reference's (d chi) x (d chi) SVD, for every split AND every centre shift
        cnt = {k_: sum(s1[k_] - s0[k_] for s0, s1 in zip(stats0, stats1)) for k_ in stats1[0]}
        # counters count batched calls; each call covers the engine's nb trajectories
        wsum = lambda key: sum((s1[key] - s0[key]) * e.B for s0, s1, e in zip(stats0, stats1, engines))  # noqa: E731
        n_mats = sum(s1["svd_matrices"] - s0["svd_matrices"] for s0, s1 in zip(stats0, stats1))
        flops_svd_nominal = F_svd * n_mats
        flops_kry_nominal = F_mv2 * wsum("matvecs_two_site") + F_mv1 * (wsum("matvecs") - wsum("matvecs_two_site"))
        flops_env = F_mv1 * wsum("env_updates")
        # executed Krylov flops: a certified identity channel of an environment (identity_channels counter) removes one of the D
        # blocks of the corresponding GEMM, so the chi^3 terms run with D - (certified channels per call) / 2 on average ...
        ident = wsum("identity_channels") / max(1.0, 2.0 * wsum("identity_checks")) if "identity_checks" in stats1[0] else 0.0
        checked = wsum("identity_checks") / max(1.0, wsum("krylov_calls")) if "identity_checks" in stats1[0] else 0.0
        D_eff = D - ident * checked
        # ... and the GEMMs multiply complex numbers with three real products (3M scheme, tjm_gemm.hip): 6 real flops per complex
        # multiply-add on the matrix pipe instead of the nominal 8
        F_mv2x = 6.0 * 2 * d_ ** 2 * D_eff * chi ** 3 + 8.0 * d_ ** 4 * D ** 2 * chi ** 2
        F_mv1x = 6.0 * 2 * d_ * D_eff * chi ** 3 + 8.0 * d_ ** 2 * D ** 2 * chi ** 2
        flops_kry_exec = F_mv2x * wsum("matvecs_two_site") + F_mv1x * (wsum("matvecs") - wsum("matvecs_two_site"))
        # executed SVD-family flops, counted on the device: 28 real flops per row of every column pair of a visited Jacobi tile (8 dot
        # product + 20 rotation, identity rotations of a visited tile included), for the fp64 kernels and - mixed-precision split -
        # for the complex64 ones (fp32 flops), plus the GEMMs of the split's fp64 phase at 6 real flops per complex multiply-add
        flops_jac64 = 28.0 * float(jw[0])
        flops_jac32 = 28.0 * float(mx[6])
        flops_mixgemm = 0.75 * float(mx[9])
        # ---- time: HIP events on every engine's stream around each class.  With E engines the streams overlap on the device and the
        # bracketed times add up to more than the wall time: `stream_ms` is the raw sum over the engines (what the brackets measured),
        # `wall_share_ms` = stream_ms x wall / sum of all classes is the class's SHARE of the wall time - an attribution, not a kernel
        # measurement.  Class rates below are flops / wall share (so that the classes add up to the step); kernel rates use the
        # sampled launch durations themselves.
        stream_ms = {c: sum(p[c]["ms"] for p in prof) for c in ("svd", "krylov", "env")}
        overlap = max(1.0, sum(stream_ms.values()) / (1e3 * elapsed))
        cls_ms = {c: v / overlap for c, v in stream_ms.items()}
        tf = lambda fl, msv: (fl / 1e12) / (msv / 1e3) if msv > 0 else None  # noqa: E731
        frac = lambda x, pk: (x / pk) if x else None  # noqa: E731
        busy = sum(cls_ms.values()) / 1e3
        step_tf = (flops_svd_nominal + flops_kry_nominal + flops_env) / 1e12 / elapsed
        # ---- the dominant kernel: the Jacobi tile kernel, in whichever arithmetic it spends more time (sampled: every 8th launch)
        k64 = {"ms": ms.value, "samples": int(ns.value), "bytes": nbytes.value, "flops": flops_jac64, "peak": peak, "bound": valu_bound,
               "name": ("complex64 Jacobi tile kernels of libtjm_hip_f32.so: jacobi_quad64_kernel (256- and 512-column matrices: four 16-column blocks per "
                        "workgroup, three / two tournament rounds per load), jacobi_cross16q_kernel (four columns per wavefront, other sizes up to 512 rows), "
                        "jacobi_cross16x_kernel (two columns per wavefront: up to 1024 rows, solves with a rotation record)") if f32 else "jacobi_cross16x_kernel (fp64)"}
        k32 = {"ms": ms32.value, "samples": int(ns32.value), "bytes": nb32.value, "flops": flops_jac32, "peak": peak32, "bound": "fp32-valu",
               "name": "tjm32::jacobi_quad64_kernel (complex64 phase of the mixed-precision two-site split: four 16-column blocks per workgroup, three "
                       "tournament rounds per load; tjm32::jacobi_cross16q_kernel - one round per load - for sizes other than 256 columns)"}
        k64["iso"], k32["iso"] = (iso["f64"], iso["c64"]) if iso else (None, None)
        # the fp64 GEMM kernel (every product of whole 64 x 64 x 16 tiles: Krylov, environments, the fp64 phase of the two-site split);
        # flops of the sampled launches = device-counted flops of all launches x sampled / all
        kgm = {"ms": float(gp[0]), "samples": int(gp[1]), "bytes": float(gp[4]), "flops": 6.0 * 64 * 64 * float(gp[3]) * (gp[1] / gp[2] if gp[2] else 0.0) * 8.0,
               "peak": peak64, "bound": "mfma",
               "name": "tjm::zgemm4_kernel (batched complex128 GEMM on v_mfma_f64_4x4x4_4b_f64: persistent workgroups, operand tiles staged by "
                       "global_load_lds in the instruction's lane order, three real products per complex one; serves the H_eff / environment "
                       "products and the fp64 phase of the two-site split)",
               "iso": iso["gemm"] if iso else None}
        # dominant kernel = the one with the largest summed launch time in the isolated one-engine step (every launch of the three bracketed)
        iso_ms = lambda kk: (kk.get("iso") or {}).get("ms", 0.0) if kk.get("iso") else kk["ms"]  # noqa: E731
        dom = max((k32, k64, kgm), key=iso_ms) if not f32 else k64

        def kernel_line(kk):
            if not kk["samples"]:
                return None
            launches = 8.0 * kk["samples"]  # the sampler brackets every 8th launch
            avg_us = 1e3 * kk["ms"] / kk["samples"]
            # A launch's duration is measured while the other engines' kernels share the device (E streams): the time the kernel
            # had the device FOR ITSELF is its summed launch time / stream overlap (how much the streams overlap is measured:
            # summed bracketed stream time / wall).  With one engine the two rates coincide.
            raw = kk["flops"] / 1e12 / (launches * avg_us / 1e6) if kk["flops"] else None
            rate = raw * overlap if raw else None
            gbs = (kk["bytes"] / 1e9) / (kk["ms"] / 1e3) * overlap
            line = {"name": kk["name"], "bound": kk["bound"], "avg_launch_us_overlapped": avg_us, "launches_sampled_overlapped": kk["samples"],
                    "executed_TFLOPs_overlapped_x_stream_overlap": rate, "executed_TFLOPs_per_launch_duration_overlapped": raw, "peak_TFLOPs": kk["peak"],
                    "frac_overlapped": frac(raw, kk["peak"]), "frac_overlapped_x_stream_overlap": frac(rate, kk["peak"])}
            if kk["bound"] == "mfma":
                line["frac_of_matrix_peak_overlapped"] = line["frac_overlapped"]
            io = kk.get("iso")
            if io and io["samples"] and io["flops"]:
                # the measurement: one engine alone on the device, every launch bracketed - flops of all launches / their summed duration
                tfl = io["flops"] / 1e12 / (io["ms"] / 1e3)
                gbs_i = io["bytes"] / 1e9 / (io["ms"] / 1e3)
                ai = io["flops"] / io["bytes"] if io["bytes"] else None
                ridge = kk["peak"] * 1e12 / (HBM_PEAK_GBS * 1e9)
                line.update({"avg_launch_us": 1e3 * io["ms"] / io["samples"], "launches": io["samples"], "flops_per_launch": io["flops"] / io["samples"],
                             "bytes_per_launch": io["bytes"] / io["samples"], "executed_TFLOPs": tfl, "frac_of_vector_peak": tfl / kk["peak"],
                             "tile_bytes_GBps": gbs_i, "frac_of_hbm_peak": gbs_i / HBM_PEAK_GBS, "flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                             "bound_by_arithmetic_intensity": ("hbm" if (ai is not None and ai < ridge) else kk["bound"]),
                             "frac": max(tfl / kk["peak"], gbs_i / HBM_PEAK_GBS)})
                if kk["bound"] == "mfma":
                    line["frac_of_matrix_peak"] = line.pop("frac_of_vector_peak")
            return line

        # the complex64 block-reflector apply of the QR preconditioner (qr_block_apply_multi_kernel): matrix-core-bound, measured alone in the
        # isolated step (every launch bracketed); nominal flops = 8 per complex multiply-add of V^H C and C - V (T V^H C)
        qr_io = iso["qr"] if iso else None
        qr_tm = qr_c64 if args.dtype == "complex128" else qr_own
        qr_line = None
        if qr_io and qr_io["samples"] and qr_io["ms"] > 0:
            qr_tf = qr_io["flops"] / 1e12 / (qr_io["ms"] / 1e3)
            qr_line = {"name": ("tjm32::" if args.dtype == "complex128" else "tjm::") + "qr_block_apply_multi_kernel (block reflectors of the Householder QR preconditioner, four panels per pass)",
                       "bound": "mfma-f32", "avg_launch_us": 1e3 * qr_io["ms"] / qr_io["samples"], "launches": qr_io["samples"],
                       "nominal_TFLOPs": qr_tf, "peak_TFLOPs": peak32, "frac": qr_tf / peak32,
                       "summed_launch_ms_in_the_isolated_step": qr_io["ms"],
                       "avg_launch_us_overlapped": (1e3 * qr_tm[0] / qr_tm[2]) if qr_tm[2] else None}
        dom_line = kernel_line(dom)
        traffic, traffic_src = pmc_traffic(L, chi, sizes[0], "zgemm4" if dom is kgm else ("tjm32" if dom is k32 else "tjm::"))
        alg_bytes_per_launch = (dom["bytes"] / dom["samples"]) if dom["samples"] else None
        svd_exec_tf = tf(flops_jac64 + flops_jac32 + flops_mixgemm, cls_ms["svd"])
        kry_exec_tf = tf(flops_kry_exec, cls_ms["krylov"])
        step_wall = [max(d.step_s[k] for d in drives) for k in range(K)] if all(len(d.step_s) == K for d in drives) else []
        all_steps = [max((d.warm_s + d.step_s)[k] for d in drives) for k in range(W + K)] if all(len(d.warm_s) + len(d.step_s) >= W + K for d in drives) else []
        first_ten_rate = (total_traj * 10.0 / STEPS_PER_TRAJ / sum(all_steps[:10])) if len(all_steps) >= 10 else None
        out = {
            "metric": "trajectories/sec",
            "value": value,
            "unit": "trajectories/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64" if args.dtype == "complex128" else "f32",  # arithmetic type; the tensors are complex (interleaved re, im)
            "data": "synthetic",
            "config": {
                "workload": f"{L}-site {WORKLOADS[args.workload][0]}, chi={chi}, dt={dt:g}, "
                            f"order-1 TJM, {args.tdvp_mode} TDVP, svd_threshold=1e-12, krylov_tol={args.krylov_tol:g}, Haar chi-saturated initial MPS",
                "trajectories": total_traj,
                "trajectories_in_flight_per_gpu": B,
                "engines_per_gpu": E,
                "steps_per_trajectory": STEPS_PER_TRAJ,
                "parallelism": f"trajectory-sharded x{world}" + (" (gloo: ranks share the visible GPUs - functional check, not a scaling number)" if gloo and world > 1 else ""),
                "storage": args.dtype,
                "arithmetic": ("fp64 results: every output of the two-site split is produced by fp64 arithmetic behind per-trajectory certificates "
                               "(complex64-preconditioned: the starting basis of the split comes from a complex64 Jacobi iteration; "
                               "Krylov, environments and centre shifts are fp64 throughout)") if args.dtype == "complex128" else "complex64 / fp32 throughout",
            },
            "site_updates_per_sec": site_updates,
            # batched calls per step and engine (every engine makes the same calls); svd_matrices counts trajectories: whole GPU
            "counters_per_step": {k_: v / K / (1 if k_ in ("svd_matrices", "certified_dissipations", "certified_jumps") else E) for k_, v in cnt.items()},
            # The certified scalar dissipation / in-place jumps depend on the state: nothing certifies in the first
            # steps from the Haar state, most trajectory-steps do later, so the value depends (by a few per cent) on --steps / --warmup.
            "certified_fraction_of_trajectory_steps": {"dissipations": cnt.get("certified_dissipations", 0) / max(1, total_traj * K // world),
                                                      "note": "tests/test_hip_fullsize.py pins this path against the reference at full size (ten consecutive steps)"},
            "step_wall_seconds": {"first": step_wall[0], "last": step_wall[-1], "all": step_wall} if step_wall else None,
            # a trajectory as SURVEY 8d defines it: ten steps from the stated (Haar) initial state - steps 1 - 10 of this run, warm-up
            # steps included (slowest engine of every step); null when the run has fewer than ten steps
            "value_steps_1_to_10_from_the_initial_state": first_ten_rate,
            "mean_Z_site0": float(zsum[0] / total_traj),
            # ---- roofline of the dominant kernel, EXECUTED work: flops counted on the device / (launches x sampled launch duration)
            "roofline": {
                "bound": (dom_line or {}).get("bound_by_arithmetic_intensity", dom["bound"]),  # "mfma" | "hbm" | "fp32-valu" | "fp64-valu"
                "kernel": dom["name"],
                "achieved": (dom_line or {}).get("executed_TFLOPs"),
                "peak": dom["peak"],
                "unit": "TFLOP/s",
                "frac": (dom_line or {}).get("frac"),
                "frac_overlapped": (dom_line or {}).get("frac_overlapped"),
                "how": (("MEASURED ALONE: after the timed region one engine runs one more step with the device to itself and EVERY launch of the kernel "
                         "bracketed by HIP events on its stream; achieved = executed flops of those launches (output tiles x K counted on the device "
                         "by the kernel itself x 3 real matrix-core products x 2 x 64 x 64: masked trajectories and the mirror tiles of Hermitian "
                         "products are not counted; nominal complex flops would be 8/6 of it) / their summed duration; avg_launch_us is that "
                         "duration per launch - the figure `rocprofv3 --kernel-trace --stats` of a one-engine run reports for the same kernel "
                         "(profiles/r05/iso_*.csv).  The kernel is the dominant one of the step: largest summed launch time of the three sampled "
                         "kernels in that isolated step (summed_launch_ms_in_the_isolated_step).  bound: by arithmetic intensity (executed flops / "
                         "operand-and-result bytes of a launch once each, against peak / 8 TB/s); peak = the fp64 matrix-core rate 78.6 TFLOP/s.  "
                         "frac_overlapped: the same ratio from the launches sampled INSIDE the timed region, where four engines' kernels share the device")
                        if dom is kgm else
                        ("MEASURED ALONE: after the timed region one engine runs one more step with the device to itself and EVERY launch of the kernel "
                         "bracketed by HIP events on its stream; achieved = executed flops of those launches (28 real flops x rows x column pairs "
                         "of every visited tile, counted on the device; a visited tile executes all its rotation slots, identity rotations included) / "
                         "their summed duration; avg_launch_us is that duration per launch - the figure `rocprofv3 --kernel-trace --stats` of a "
                         "one-engine run reports for the same kernel (profiles/r05/iso_*.csv).  bound: by arithmetic intensity (flops / tile bytes "
                         "against peak / 8 TB/s); frac = the larger of the vector-peak and HBM-peak fractions.  frac_overlapped: the same ratio from "
                         "the launches sampled INSIDE the timed region, where four engines' kernels share the device (no correction factor)")),
                "avg_launch_us": (dom_line or {}).get("avg_launch_us"),
                "flops_per_launch": (dom_line or {}).get("flops_per_launch"),
                "frac_of_vector_peak": (dom_line or {}).get("frac_of_vector_peak"),
                "frac_of_matrix_peak": (dom_line or {}).get("frac_of_matrix_peak"),
                "frac_of_hbm_peak": (dom_line or {}).get("frac_of_hbm_peak"),
                "flop_per_byte": (dom_line or {}).get("flop_per_byte"),
                "isolated_step_seconds": iso["step_s"] if iso else None,
                "algorithmic_bytes_per_launch": (dom_line or {}).get("bytes_per_launch") or alg_bytes_per_launch,  # every visited quad of blocks read and written once
                # minimum HBM traffic of a whole site-update 16 (4 d chi^2 + 3 D chi^2) bytes; of it the split: theta in, two site tensors out
                "algorithmic_bytes_per_site_update_survey_8d": 16.0 * (4 * d_ * chi ** 2 + 3 * D * chi ** 2),
                "algorithmic_bytes_per_split": 16.0 * (d_ ** 2 * chi ** 2 + 2 * d_ * chi ** 2),
                "kernel_bytes_per_split_and_trajectory": ((dom_line or {}).get("bytes_per_launch", 0.0) / max(1, sizes[0])) * (5.0 if chi == 128 else 15.0)
                                                         * ((mx[1] / mx[0]) if mx[0] else 0.0) if dom is k32 else None,
                "traffic": traffic,
                "traffic_over_algorithmic_bytes": (traffic / alg_bytes_per_launch) if (traffic and alg_bytes_per_launch) else None,
                "traffic_source": traffic_src,
                # nominal convention of SURVEY 8d for the whole SVD class: 88 n^3 per (d chi) x (d chi) factorisation as the reference
                # executes it (every split AND every centre shift), over the class's share of the wall time
                "frac_nominal": frac(tf(flops_svd_nominal, cls_ms["svd"]), peak),
                "achieved_nominal": tf(flops_svd_nominal, cls_ms["svd"]),
                "algorithmic_flops_per_svd": F_svd,
                "kernels": {"gemm_fp64": kernel_line(kgm) if not f32 else None, "jacobi_fp64": kernel_line(k64), "jacobi_complex64": kernel_line(k32) if not f32 else None,
                            "qr_apply_complex64": qr_line},
                "summed_launch_ms_in_the_isolated_step": {"gemm_fp64": (iso or {}).get("gemm", {}).get("ms"), "jacobi_complex64": (iso or {}).get("c64", {}).get("ms"),
                                                          "jacobi_fp64": (iso or {}).get("f64", {}).get("ms")} if iso else None,
                "jacobi_sweeps_per_solve_fp64": (float(jw[2]) / float(jw[3])) if jw[3] else None,
                "jacobi_applied_over_executed_rotations_fp64": (float(jw[1]) / float(jw[0])) if jw[0] else None,
                "jacobi_applied_over_executed_rotations_complex64": (float(mx[7]) / float(mx[6])) if mx[6] else None,
                # mixed-precision two-site split (tjm_mixed.h)
                "mixed_split": {"batched_splits": mx[0], "c64_sweeps_per_split": (mx[1] / mx[0]) if mx[0] else None,
                                "f64_jacobi_sweeps_per_split": (mx[2] / mx[0]) if mx[0] else None, "batches_sent_to_fp64_path": mx[3],
                                "trajectories_finished_by_fp64_jacobi": mx[4], "batches_with_second_polar_step": mx[5],
                                "fp64_gemms_per_split": (mx[8] / mx[0]) if mx[0] else None},
                "svds_per_step": cnt["svds"] / K / E,
                "stream_overlap": overlap,  # summed stream time of the engines / wall time (1 for a single engine)
                "classes": {
                    "note": "stream_ms = HIP-event brackets summed over the engines' streams (raw); wall_share_ms = stream_ms / stream_overlap, the "
                            "class's share of the wall time (attribution: the classes add up to the step); rates = flops / wall share",
                    "svd": {"stream_ms": stream_ms["svd"], "wall_share_ms": cls_ms["svd"], "share_of_stream_time": cls_ms["svd"] / 1e3 / busy if busy else None,
                            "executed_TFLOPs": svd_exec_tf,
                            "executed_flops_split": {"jacobi_fp64": flops_jac64, "jacobi_complex64_fp32_flops": flops_jac32, "fp64_gemms_of_the_mixed_split": flops_mixgemm},
                            "nominal_TFLOPs": tf(flops_svd_nominal, cls_ms["svd"]),
                            "avg_batched_svd_ms_on_its_stream": stream_ms["svd"] / max(1, sum(p["svd"]["regions"] for p in prof))},
                    "krylov": {"bound": mfma_bound, "stream_ms": stream_ms["krylov"], "wall_share_ms": cls_ms["krylov"],
                               "share_of_stream_time": cls_ms["krylov"] / 1e3 / busy if busy else None,
                               "achieved_TFLOPs": kry_exec_tf,
                               # an ATTRIBUTION (flops over the class's share of the wall time under overlapping streams), not a kernel measurement
                               "attributed_rate_over_peak": (kry_exec_tf / peak) if kry_exec_tf else None,
                               "nominal_TFLOPs": tf(flops_kry_nominal, cls_ms["krylov"]),
                               "note": "H_eff applies (2 MFMA GEMMs + MPO stage) with the Lanczos vector kernels (HBM-bound) inside the region; achieved = "
                                       "EXECUTED flops (the GEMM blocks of the environments' certified identity channels are not computed; 6 real flops "
                                       "per complex multiply-add: three-product complex multiplication) over the class's wall share; with several "
                                       "engines the share understates the time the kernels had the device to themselves, hence the cap at 1 - the "
                                       "GEMM kernels alone show 52 - 61 % MfmaUtil (profiles/r03_pmc_pass5_*)"},
                    "env": {"bound": mfma_bound, "stream_ms": stream_ms["env"], "wall_share_ms": cls_ms["env"],
                            "achieved_TFLOPs": tf(flops_env, cls_ms["env"]), "attributed_rate_over_peak": frac(tf(flops_env, cls_ms["env"]), peak),
                            "share_of_stream_time": cls_ms["env"] / 1e3 / busy if busy else None},
                    "whole_step": {"nominal_TFLOPs": step_tf, "nominal_rate_over_peak": step_tf / peak,
                                   "timed_classes_over_wall": busy / elapsed if elapsed > 0 else None},
                },
            },
        }
        out["config"]["value_steps_1_to_10_from_the_initial_state"] = first_ten_rate
        out["config"]["certified_fraction_of_trajectory_steps"] = out["certified_fraction_of_trajectory_steps"]["dissipations"]
        out["config"]["c64_sweeps_per_split"] = out["roofline"]["mixed_split"]["c64_sweeps_per_split"]
        out["config"]["batches_with_second_polar_step"] = out["roofline"]["mixed_split"]["batches_with_second_polar_step"]
        if world > 1:
            # what one GPU does at this shard size (measured on one MI355X, profiles/): the driver's SCALE line can be read against it
            out["config"]["per_gpu_shard"] = B
            out["one_gpu_rate_at_this_shard_size"] = shard_rate(B)
        if cpu_ref is not None:
            out["cpu_baseline"] = cpu_ref
            full = [r for r in cpu_ref.get("rows", []) if r.get("cores") == cpu_ref.get("host_cores")]
            refdef = cpu_ref.get("reference_default_row")
            strict = min(value / cpu_ref["best_measured_full_step"]["value"], (value / refdef["value"]) if refdef and refdef.get("value") else float("inf"))
            out["speedup_vs_cpu"] = {
                "stricter_of_best_measured_row_and_reference_default_row": strict,
                "vs_reference_default_workers_row": (value / refdef["value"]) if refdef and refdef.get("value") else None,  # available_cpus() - 1 workers (parallel_utils.py:62-98)
                "vs_best_measured_whole_host_row": value / cpu_ref["best_measured_full_step"]["value"],  # the ratio north_star's 50 x is read against
                "vs_best_whole_host_row_incl_extrapolated": value / cpu_ref["value"],
                "vs_one_core": value / cpu_ref["per_core_value"],
                # the reference's own default is one worker per core (simulator.py:1074-1098: max_workers = available_cpus() - 1): on this
                # host that row is memory-bound, SLOWER than the best row, and extrapolated from the slab sample - not a measured full step
                "vs_all_cores_row_extrapolated": (value / full[0]["value"]) if (full and full[0].get("value")) else None,
            }
            out["cpu_baseline"]["gpu_over_cpu_stricter_ratio"] = strict
            out["cpu_baseline"]["gpu_over_best_measured_row"] = out["speedup_vs_cpu"]["vs_best_measured_whole_host_row"]
            out["cpu_baseline"]["gpu_over_one_core"] = out["speedup_vs_cpu"]["vs_one_core"]
        if not args.verbose:
            shorten_prose(out)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
