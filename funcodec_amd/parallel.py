"""Multi-GPU layout of the hot path: one process per GPU, utterances sharded contiguously, no
collective inside the data path, one RCCL all-gather of the code indices at the end
(SURVEY.md §8e).  The reference's own scheme is N independent processes + `cat`
(egs/LibriTTS/codec/encoding_decoding.sh:59-101).
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `n_items` utterances owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_codes(codes: torch.Tensor, dist=None, shard_sizes=None) -> torch.Tensor:
    """All-gather code indices [n_q, B_local, Tf] -> [n_q, B_global, Tf] in rank order.

    Ranks may hold different B_local (ragged shards): sizes are exchanged first and shards are padded to
    the largest one, so a single `all_gather_into_tensor` (one RCCL ring over xGMI) moves the payload.
    `shard_sizes` (every rank's B_local, e.g. from `shard_range`) skips the size exchange and with it the only
    host synchronisation of the call, so the gather stays asynchronous on the stream."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return codes
    return _gather_ranks(codes, dist, shard_sizes)


def _gather_ranks(codes: torch.Tensor, dist, shard_sizes=None) -> torch.Tensor:
    """The collective itself, for any world size >= 1 (a 1-rank communicator takes exactly the N-rank code path: this is what the
    hardware smoke test on the 1-GPU boxes runs, tests/test_multigpu_gpu.py)."""
    world = dist.get_world_size()
    n_q, b_local, tf = codes.shape
    if shard_sizes is not None:
        all_sizes = [int(v) for v in shard_sizes]
        assert len(all_sizes) == world and all_sizes[dist.get_rank()] == b_local
    else:
        sizes = torch.tensor([b_local], dtype=torch.int64, device=codes.device)
        all_sizes = torch.empty(world, dtype=torch.int64, device=codes.device)
        dist.all_gather_into_tensor(all_sizes, sizes)
        all_sizes = all_sizes.tolist()
    b_max = max(all_sizes)
    send = codes.permute(1, 0, 2).contiguous()                 # [B_local, n_q, Tf]: batch-major for concatenation
    if b_local < b_max:
        pad = torch.zeros((b_max - b_local, n_q, tf), dtype=codes.dtype, device=codes.device)
        send = torch.cat([send, pad], 0)
    recv = torch.empty((world * b_max, n_q, tf), dtype=codes.dtype, device=codes.device)
    dist.all_gather_into_tensor(recv, send)
    parts = [recv[r * b_max: r * b_max + all_sizes[r]] for r in range(world)]
    return torch.cat(parts, 0).permute(1, 0, 2).contiguous()
