"""world_size-2 (gloo, CPU) test of the multi-GPU path of the particle-filter frame loop.

What is under test is the sharding / exchange logic of pf.ParticleFilter (all-reduce of the weight
normaliser, all-gather of shard totals and of offspring offsets, all-to-all migration of particles
and their landmark maps).  The arithmetic stages are supplied by the CPU checker (tests/_oracle_ops.py)
because there is no GPU here; on the GPU the same class runs with HipOps (tests/test_gpu_pf.py).
Property: the sharded run equals the unsharded run bit for bit, for poses, maps and log-weights.
"""
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import _shard_worker as W


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("L,world", [(6, 2), (0, 2), (6, 4), (0, 3), (6, 8)])
def test_sharded_equals_unsharded(orc, tmp_path, L, world):
    n_total, frames = 768, 6
    ref = W.run_filter(0, 1, n_total, L, frames)
    mp.spawn(W.worker, args=(world, _free_port(), n_total, L, frames, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    pose = np.concatenate([p["pose"] for p in parts], axis=1)
    assert np.array_equal(pose.view(np.uint32), ref["pose"].view(np.uint32))
    logw = np.concatenate([p["logw"] for p in parts])
    assert np.array_equal(logw.view(np.uint32), ref["logw"].view(np.uint32))
    if L:
        mapc = np.concatenate([p["map"] for p in parts], axis=0)   # rows = particles
        assert np.array_equal(mapc.view(np.uint32), ref["map"].view(np.uint32))
    # the scenario really exercised the exchange: the upper half of the population collapses and is refilled
    # from the lower half, i.e. from other ranks (several sources per receiver when world > 2)
    assert parts[-1]["migrated"].max() > 10   # rows, i.e. distinct ancestors (each travels once per destination)
    # every rank agrees on the heaviest particle, and it is the unsharded answer
    for p in parts:
        assert tuple(p["best"]) == tuple(np.array(ref["best"]))
