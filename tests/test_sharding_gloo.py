"""N>1 path on CPU: two processes over gloo run the clip-sharded sampler (the per-rank compute is the
oracle at a tiny size - tests may use it) and must reproduce the unsharded result bit for bit, for even
and ragged shard sizes."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, batch_noise, oracle_params, xf_pair


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sample(xfp, xfo, noise, S):
    p = oracle_params()
    torch.set_num_threads(1)
    with torch.no_grad():
        return O.ddim_sample_loop(p, noise, xfp, xfo, [noise.shape[1]] * noise.shape[0], S)


def _worker(rank, world, port, B, T, S, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_conductor_amd.sharding import shard_bounds, sharded_sample
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    lo, hi = shard_bounds(B, rank, world)

    def fn(mel_shard, noise_shard):     # `mel` here is just the carrier of the clip axis
        assert mel_shard.shape[0] == hi - lo
        return _sample(xfp[lo:hi], xfo[lo:hi], noise_shard, S)

    full = sharded_sample(fn, torch.zeros(B, 3, 1), noise)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(B, tmp_path):
    T, S, world = 64, 25, 2   # linear schedule needs S > 20 (beta_end = 20/S must stay < 1)
    out = str(tmp_path / f"gathered_{B}.npy")
    mp.spawn(_worker, args=(world, _free_port(), B, T, S, out), nprocs=world, join=True)
    from diffusion_conductor_amd.sharding import shard_bounds
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    # the same shards computed in this process, concatenated: gathering must not change a bit
    parts = [_sample(xfp[lo:hi], xfo[lo:hi], noise[lo:hi], S) for lo, hi in (shard_bounds(B, r, world) for r in range(world))]
    got = np.load(out)
    assert got.shape == (B, T, 26)
    assert np.array_equal(got, torch.cat(parts).numpy())
    # and the joint batch agrees to rounding (CPU GEMM blocking depends on the batch size)
    joint = _sample(xfp, xfo, noise, S).numpy()
    assert np.linalg.norm(got - joint) <= 1e-5 * np.linalg.norm(joint)


def test_two_rank_even_shards(tmp_path):
    _run(4, tmp_path)


def test_two_rank_ragged_shards(tmp_path):
    _run(3, tmp_path)


def _worker_small(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_conductor_amd.sharding import shard_bounds, sharded_sample
    B, T = 1, 8
    calls = []

    def fn(mel_shard, noise_shard):
        calls.append(mel_shard.shape[0])
        return noise_shard * 2.0

    noise = torch.arange(B * T * 26, dtype=torch.float32).reshape(B, T, 26)
    full = sharded_sample(fn, torch.zeros(B, 3, 1), noise)
    lo, hi = shard_bounds(B, rank, world)
    assert calls == ([1] if hi > lo else [])                 # the rank with the empty shard never calls the sampler ...
    assert torch.equal(full, noise * 2.0)                     # ... but takes part in the gather and gets the result
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_fewer_clips_than_ranks(tmp_path):
    """B=1 on 2 ranks (the last batch of a dataset): rank 1's shard is empty - it must still enter the all-gather."""
    out = str(tmp_path / "small.npy")
    mp.spawn(_worker_small, args=(2, _free_port(), out), nprocs=2, join=True)
    assert np.load(out).shape == (1, 8, 26)


def _worker_unsized(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffusion_conductor_amd.sharding import sharded_sample
    msg = "no error"
    try:
        sharded_sample(lambda m, n: torch.zeros(m.shape[0], 8, 26), torch.zeros(1, 3, 1), None)      # B=1 < world, nothing sizes the empty shard
    except ValueError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(msg)
    dist.barrier()                                            # reachable: NO rank is stuck in the all-gather
    dist.destroy_process_group()


def test_unsized_empty_shard_raises_on_every_rank(tmp_path):
    """Fewer clips than ranks with neither out_shape nor noise: the argument error is raised on EVERY rank before anyone samples -
    raised only on the rank with the empty shard it would leave the others waiting in the collective (round 2's advisor finding)."""
    mp.spawn(_worker_unsized, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "fewer clips than ranks" in (tmp_path / f"rank{r}.txt").read_text()


def _worker_evidence(rank, world, port, out_dir, fault):
    """bench.py's multi_gpu_evidence (the N > 1 branch of the bench line) on two gloo ranks with stub samplers: the identity
    de-duplication, the per-rank time gather, gather_block_equals_local and sharded_equals_single_gpu."""
    import json
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from diffusion_conductor_amd.sharding import gather_poses
    B, T, P = 3, 5, 26

    def shard(r):                     # any deterministic per-rank "sampling result"
        return (torch.arange(B * T * P, dtype=torch.float32).reshape(B, T, P) + 1000.0 * r).sin()

    local = shard(rank)
    gathered = gather_poses(local, world * B)
    if fault == "block" and rank == 1:
        gathered = gathered.clone()
        gathered[rank * B, 0, 0] += 1.0          # this rank's block of the gather differs from what it sampled
    ident = "same device" if fault == "ident" else f"device {rank}"
    res = None
    try:
        res = bench.multi_gpu_evidence(dist, torch.device("cpu"), ident, rank, world, B, 0.010 * (rank + 1), gathered,
                                       lambda: shard(rank), shard)
    except AssertionError as e:
        res = {"assertion": str(e)}
    with open(os.path.join(out_dir, f"ev{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_gpu_evidence_on_two_gloo_ranks(tmp_path):
    import json
    for fault in ("none", "block", "ident"):
        d = tmp_path / fault
        d.mkdir()
        mp.spawn(_worker_evidence, args=(2, _free_port(), str(d), fault), nprocs=2, join=True)
        ev = [json.load(open(d / f"ev{r}.json")) for r in range(2)]
        if fault == "ident":             # two ranks on one device: refused on every rank (before any collective a rank could hang in)
            assert all("ranks share a device" in e["assertion"] for e in ev)
            continue
        for r, e in enumerate(ev):
            assert e["rccl_ranks"] == 2 and e["backend"] == "gloo" and e["devices"] == ["device 0", "device 1"]
            assert e["ms_per_step_by_rank"] == {"min": 10.0, "max": 20.0}
            assert e["gather_block_equals_local"] is (fault == "none")           # the MIN over ranks: one bad block fails every rank's line
            assert e["all_gather_bytes_per_rank"] == 3 * 5 * 26 * 4
            assert e["sharded_equals_single_gpu"] is (True if r == 0 else None)  # rank 0 re-samples the last rank's shard (its block is intact on rank 0)
