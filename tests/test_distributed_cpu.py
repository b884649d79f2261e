"""World-size-2 gloo test of the trajectory sharding + the final all-reduce (the only collective of the path)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from yaqs_amd.tjm import gather_counts, gather_trajectories, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    num_traj, n_obs, T = 7, 3, 4
    lo, hi = shard_range(num_traj, rank, world)
    # every trajectory row is a pure function of its index (as in the real path)
    res = np.stack([np.full((n_obs, T), float(t)) + np.arange(T) for t in range(lo, hi)]) if hi > lo else np.zeros((0, n_obs, T))
    diag = np.stack([np.full((3, T), 10.0 * t) for t in range(lo, hi)]) if hi > lo else np.zeros((0, 3, T))
    full_r, full_d = gather_trajectories(res, diag, num_traj, lo, "cpu")
    # measurement histograms of the circuit path: basis states are L-bit Python integers, ranks hold different (possibly no) keys
    mine = {0: {5: 2, (1 << 70) + 3: 1}, 1: {5: 4, 9: 7}}[rank]
    total = gather_counts(mine, "cpu")
    assert total == {5: 6, (1 << 70) + 3: 1, 9: 7}, total
    assert gather_counts({}, "cpu") == {}
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full_r)
    np.save(os.path.join(out_dir, f"d{rank}.npy"), full_d)
    dist.destroy_process_group()


def test_sharded_trajectories_allreduce_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    expect_r = np.stack([np.full((3, 4), float(t)) + np.arange(4) for t in range(7)])
    expect_d = np.stack([np.full((3, 4), 10.0 * t) for t in range(7)])
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"r{rank}.npy"), expect_r)
        assert np.array_equal(np.load(tmp_path / f"d{rank}.npy"), expect_d)


def _gpu_worker(rank, world, port, out_dir):
    """Two gloo ranks on the one GPU of the test box: the sharding, the all-reduce of the rows and the histogram gather of
    Simulator.run / run_circuit end to end (the collectives run on CPU tensors under gloo, the trajectories on the GPU)."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, cres = _sharded_case()
    np.save(os.path.join(out_dir, f"a{rank}.npy"), np.stack(res.trajectories))
    np.save(os.path.join(out_dir, f"c{rank}.npy"), np.stack(cres.trajectories))
    np.save(os.path.join(out_dir, f"k{rank}.npy"), np.array(sorted(cres.counts.items()), dtype=np.int64))
    dist.destroy_process_group()


def _sharded_case():
    from yaqs_amd.api import AnalogSimParams, DigitalSimParams, MPO, MPS, NoiseModel, Observable, Z, ising_trotter_layers
    from yaqs_amd.tjm import Simulator

    L = 6
    noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": 0.2} for i in range(L)])
    obs = [Observable(Z(), s) for s in range(L)]
    p = AnalogSimParams(observables=obs, elapsed_time=0.3, dt=0.1, num_traj=7, max_bond_dim=8, svd_threshold=1e-10, random_seed=4)
    sim = Simulator(device="cuda:0")
    res = sim.run(MPS(L, state="x+"), MPO.ising(L, 1.0, 0.5), p, noise)
    dp = DigitalSimParams(observables=obs, num_traj=5, shots=35, max_bond_dim=8, svd_threshold=1e-10, random_seed=4)
    cres = sim.run_circuit(MPS(L, state="zeros"), ising_trotter_layers(L, 1.0, 0.5, 0.1, 3), dp, noise)
    return res, cres


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_share_the_trajectories_on_the_gpu(tmp_path):
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res, cres = _sharded_case()  # the same runs in one process
    for rank in range(world):
        assert np.allclose(np.load(tmp_path / f"a{rank}.npy"), np.stack(res.trajectories), atol=1e-12)
        assert np.allclose(np.load(tmp_path / f"c{rank}.npy"), np.stack(cres.trajectories), atol=1e-12)
        assert np.array_equal(np.load(tmp_path / f"k{rank}.npy"), np.array(sorted(cres.counts.items()), dtype=np.int64))
    assert sum(cres.counts.values()) == 35


# ---- bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run) -----------------------------------
_RANK_SCRIPT = '''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
mode = sys.argv[1]
if mode == "ok":
    import torch.distributed as dist
    import torch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"sum": t.item(), "world": world, "argv": sys.argv[1:]}))
    sys.exit(0)
if mode == "die" and rank == 1:
    sys.exit(7)            # a rank >= 1 that ends before the rendezvous
if mode == "die":
    time.sleep(120)        # rank 0 would sit in the rendezvous / the collective's timeout
'''


def test_bench_launcher_starts_ranks_and_relays_rank0(tmp_path, capsys):
    import bench

    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = bench.launch_ranks(2, ["ok", "--steps", "3"], script=str(script))
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0
    out = [ln for ln in out if not ln.startswith("[Gloo]")]  # gloo's own connection banner goes to stdout too
    assert len(out) == 1  # exactly rank 0's line
    import json

    rec = json.loads(out[0])
    assert rec == {"sum": 3.0, "world": 2, "argv": ["ok", "--steps", "3"]}


def test_bench_launcher_does_not_hang_when_a_rank_dies(tmp_path):
    import time

    import bench

    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    t0 = time.time()
    rc = bench.launch_ranks(3, ["die"], script=str(script))
    assert rc == 7                      # the failing rank's code is the launcher's
    assert time.time() - t0 < 30.0      # and the sleeping ranks were stopped, not waited for


@pytest.mark.gpu
def test_bench_with_two_ranks_on_one_gpu_reproduces_the_single_rank_ensemble():
    """`bench.py --gpus 2 --dist-backend gloo`: the launcher, the contiguous sharding, the all-reduce of the observable sums and the
    max-over-ranks timing end to end on a box with ONE GPU (the ranks share it; a functional check, not a scaling number).  The
    ensemble mean must be the single-rank one bit for bit: trajectories are pure functions of (seed, index)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, os.path.join(root, "bench.py"), "--length", "8", "--chi", "8", "--trajectories", "8", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--engines", "2"]
    two = subprocess.run(common + ["--gpus", "2", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    one = subprocess.run(common, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert a["n_gpus"] == 2 and a["scaling"] == "strong" and a["config"]["trajectories"] == 8 and a["config"]["trajectories_in_flight_per_gpu"] == 4
    assert b["n_gpus"] == 1
    assert a["mean_Z_site0"] == b["mean_Z_site0"]
