"""Tuning probe (not product code): do two half-batches on two HIP streams overlap usefully (the latency-bound persistent LSTM of
one half under the MFMA-bound convs of the other)?  Two engine instances (own workspaces), 8 + 8 utterances vs one 16-utterance call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from funcodec_amd.config import arch_from_config, recipe_config
from funcodec_amd.model import EncodecMI355X
from funcodec_amd.synth import make_state_dict, synthetic_audio

arch = arch_from_config(recipe_config("ds640"))
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(arch, 0).items()}
engs = []
for _ in range(2):
    m = EncodecMI355X(arch, "cuda:0"); m.load_state_dict(sd); engs.append(m.engine)
wav = torch.from_numpy(synthetic_audio(16, 160000, 1234)).cuda()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
split = int(os.environ.get("SPLIT", "8"))
halves = [wav[:split].contiguous(), wav[split:].contiguous()]

def one():
    return engs[0].encode_decode(wav, 32)

def two():
    outs = []
    for e, s, h in zip(engs, streams, halves):
        with torch.cuda.stream(s):
            outs.append(e.encode_decode(h, 32))
    return outs

ref = one()
for fn, name in ((one, "one call, 16 utterances"), (two, f"two streams, {split}+{16-split}")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): r = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 100:.3f} ms per 16 utterances")
outs = two(); torch.cuda.synchronize()
codes = torch.cat([o["codes"] for o in outs], 1)
print("codes identical:", torch.equal(codes, ref["codes"]))
for e in engs: e.check_status()
