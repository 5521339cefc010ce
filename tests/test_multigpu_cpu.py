"""The N > 1 path on CPU: world_size-2 gloo processes run the same sharding + gather the GPU ranks run."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from funcodec_amd.parallel import gather_codes, shard_range


def _worker(rank, world, port, total, n_q, tf, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, rank, world)
        # "codes" of utterance u are a pure function of u, so the gathered tensor is checkable
        u = torch.arange(lo, hi, dtype=torch.int64)
        codes = (u[None, :, None] * 1000 + torch.arange(n_q)[:, None, None] * 10 + torch.arange(tf)[None, None, :]).contiguous()
        full = gather_codes(codes, dist)
        # the same with the shard sizes known up front (no size exchange, no host sync): what bench.py does
        sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
        assert torch.equal(gather_codes(codes, dist, shard_sizes=sizes), full)
        exp_u = torch.arange(total, dtype=torch.int64)
        expect = exp_u[None, :, None] * 1000 + torch.arange(n_q)[:, None, None] * 10 + torch.arange(tf)[None, None, :]
        ok = torch.equal(full, expect)
        if rank == 0:
            with open(out, "wt") as f:
                f.write("ok" if ok else "mismatch")
        assert ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5, 1])     # even shards, ragged shards, one rank empty
def test_gather_codes_world2_gloo(tmp_path, total):
    out = str(tmp_path / "r")
    port = 29500 + (os.getpid() % 2000) + total
    mp.spawn(_worker, args=(2, port, total, 4, 6, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_gather_codes_single_process_is_identity():
    x = torch.arange(24).reshape(2, 3, 4)
    assert gather_codes(x, None) is x
