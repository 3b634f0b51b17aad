"""Clip-level data parallelism: one process per GPU, contiguous clip shards, no data-path
collective except ONE all-gather of the final poses.

Clips are independent in eval mode (BatchNorm uses running stats, LayerNorm is per token, both
softmaxes stay inside a clip), so shards never exchange anything during the DDIM loop.
`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_clips: int, rank: int, world: int):
    """Contiguous [lo, hi) block of clips for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_clips, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def dist_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def gather_poses(local: torch.Tensor, n_clips: int, group=None) -> torch.Tensor:
    """All-gather per-rank results [b_local, T, P] into [n_clips, T, P] on every rank.
    Ragged shards are padded to the largest shard for the collective."""
    rank, world = dist_info(group)
    if not (dist.is_available() and dist.is_initialized()):
        return local
    sizes = [shard_bounds(n_clips, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < bmax:
        pad = torch.cat([local, local.new_zeros((bmax - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    out = local.new_empty((world * bmax,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    return torch.cat([out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sharded_sample(sample_fn, mel, noise=None, group=None, out_shape=None, device=None):
    """Run `sample_fn(mel_shard, noise_shard) -> [b_local,T,P]` on this rank's clips and gather.
    mel: [B, Tm, 128], on the device or still on the host (each rank then copies only its own clips; `device` = where the
    results live, default mel.device); noise: [B, T, P] or None.  A rank whose shard is empty (fewer clips than ranks, e.g. the
    last batch of a dataset) samples nothing and contributes zero rows of `out_shape` = (T, P) to the gather -
    it must still enter the collective, or the other ranks would wait for it forever."""
    rank, world = dist_info(group)
    B = mel.shape[0]
    # validated on EVERY rank before anyone samples: raising on the one rank whose shard is empty would leave the others
    # waiting in the all-gather
    if B < world and out_shape is None and noise is None:
        raise ValueError("sharded_sample: fewer clips than ranks needs out_shape=(T, P) or noise to size the empty shards' contribution")
    lo, hi = shard_bounds(B, rank, world)
    if hi > lo:
        local = sample_fn(mel[lo:hi], None if noise is None else noise[lo:hi])
    else:
        if out_shape is None:
            out_shape = tuple(noise.shape[1:])
        local = torch.zeros((0,) + tuple(out_shape), dtype=torch.float32, device=mel.device if device is None else device)
    return gather_poses(local, B, group)
