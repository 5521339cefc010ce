"""Worker of tests/test_multigpu_gpu.py: one process per GPU (torchrun), utterances sharded, codes all-gathered over RCCL."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.model import EncodecMI355X
    from funcodec_amd.parallel import gather_codes, shard_range
    from funcodec_amd.synth import make_state_dict, synthetic_audio
    arch = arch_from_config(recipe_config("ds320"))
    model = EncodecMI355X(arch, f"cuda:{local}")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(arch, 0).items()})
    total = 5                                                      # ragged shards: 3 + 2
    wav_all = torch.from_numpy(synthetic_audio(total, 32000, 77, "tones"))
    lo, hi = shard_range(total, rank, world)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    wav = wav_all[lo:hi].cuda()
    # a concurrent RCCL kernel on a side stream while the persistent LSTM (grid barrier, one workgroup per CU) runs
    side = torch.cuda.Stream()
    junk = torch.ones(1 << 22, device="cuda")
    for it in range(3):
        with torch.cuda.stream(side):
            for _ in range(4):
                dist.all_reduce(junk)
        r = model.engine.encode_decode(wav, 32)
        codes = gather_codes(r["codes"], dist, shard_sizes=sizes)
    torch.cuda.synchronize()
    model.engine.check_status()                                    # no barrier timeout went unnoticed
    assert codes.shape == (32, total, 100)
    # one step shaped like bench.py's Config C path (BASELINE.json configs[2]): the ds640 net, this rank's utterances walked in
    # micro-batches of 32 like bench.py (the persistent LSTM runs the second 16-utterance tile as a second launch), checked against
    # calls of 16, the gather inside the step; scaled down to 40 utterances of 1 s per rank so that the first >= 2-GPU box to run
    # the suite exercises the exact bench code path
    arch_c = arch_from_config(recipe_config("ds640"))
    model_c = EncodecMI355X(arch_c, f"cuda:{local}")
    model_c.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(arch_c, 0).items()})
    model_c.engine.micro_batch = 32
    per = 40
    wav_c = torch.from_numpy(synthetic_audio(per, 16000, 1234 + rank)).cuda()
    parts = [model_c.engine.encode_decode(wav_c[i:i + 32], 32, use_scale=True)["codes"] for i in range(0, per, 32)]
    halves = [model_c.engine.encode_decode(wav_c[i:i + 16], 32, use_scale=True)["codes"] for i in (0, 16)]
    assert torch.equal(parts[0], torch.cat(halves, 1)), "a 32-utterance call differs from two 16-utterance calls"
    codes_c = gather_codes(torch.cat(parts, 1), dist, shard_sizes=[per] * world)
    torch.cuda.synchronize()
    model_c.engine.check_status()
    assert codes_c.shape == (32, per * world, 25), codes_c.shape
    assert torch.equal(codes_c[:, rank * per:(rank + 1) * per], torch.cat(parts, 1))        # this rank's shard sits at its rank offset
    if rank == 0:
        single = model.engine.encode_decode(wav_all.cuda(), 32)    # the whole batch on ONE GPU
        assert torch.equal(single["codes"], codes), "gathered codes differ from the single-GPU result"
        assert torch.equal(single["recon"][lo:hi], r["recon"])
        with open(sys.argv[1], "wt") as f:
            f.write(f"ok ranks_seen={dist.get_world_size()}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
