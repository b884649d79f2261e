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

    from yaqs_amd.tjm import gather_trajectories, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    num_traj, n_obs, T = 7, 3, 4
    lo, hi = shard_range(num_traj, rank, world)
    # every trajectory row is a pure function of its index (as in the real path)
    res = np.stack([np.full((n_obs, T), float(t)) + np.arange(T) for t in range(lo, hi)]) if hi > lo else np.zeros((0, n_obs, T))
    diag = np.stack([np.full((3, T), 10.0 * t) for t in range(lo, hi)]) if hi > lo else np.zeros((0, 3, T))
    full_r, full_d = gather_trajectories(res, diag, num_traj, lo, "cpu")
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
