"""CPU, world_size 2 over gloo: the N>1 path of bench.py shards frames per rank with no
data-path collective and gathers the per-step 12-vectors once at the end."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import harmonic_energy as o_he
    # each rank owns its own (small) batch: seed differs per rank exactly like bench.py
    frames = 16
    x = bench.synth_signal(20260101 + rank, frames=frames)
    assert x.shape[0] == (frames - 1) * bench.HOP + bench.N_FFT
    steps = 3
    mine = np.stack([o_he.he_compute(x, bench.FS, bench.N_FFT, bench.HOP) for _ in range(steps)])  # stand-in for the HIP step
    t = torch.from_numpy(mine)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.barrier()
    dist.all_gather(gathered, t)                       # the one collective of the job
    el = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)          # max-over-ranks timing
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), torch.stack(gathered).numpy())
    assert abs(float(el) - 0.1 * world) < 1e-12
    dist.destroy_process_group()


def test_two_ranks_shard_and_gather(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert a.shape == (2, 3, 12)
    np.testing.assert_array_equal(a, b)                # every rank holds every rank's 12-vectors
    assert not np.allclose(a[0], a[1])                 # ranks processed different batches
    assert np.array_equal(a[0][0], a[0][1])            # steps are deterministic
