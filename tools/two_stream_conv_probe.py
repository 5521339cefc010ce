"""Tuning probe (round 5, not product code): does running the CONV layers of two half-batches on two HIP streams hide the ~20 - 30 us of
start-up + drain every conv launch carries?  The decoder's conv layers through fc_layer_forward: one
engine / one stream / 16 utterances against two engines / two streams / 8 + 8 utterances (launches interleaved).  No LSTM involved (round 2's
two-stream probe lost 3 ms to the latency-bound recurrence running twice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funcodec_amd.config import arch_from_config, recipe_config
from funcodec_amd.model import EncodecMI355X
from funcodec_amd.synth import make_state_dict

arch = arch_from_config(recipe_config("ds640"))
sd = {k: torch.from_numpy(v) for k, v in make_state_dict(arch, 0).items()}
engs = []
for _ in range(2):
    m = EncodecMI355X(arch, "cuda:0"); m.load_state_dict(sd); engs.append(m.engine)
LAYERS = [("decoder.model.3.convtr", 1024, 250), ("decoder.model.4.shortcut.conv", 512, 2000), ("decoder.model.4.block.1.conv", 512, 2000),
          ("decoder.model.4.block.3.conv", 256, 2000), ("decoder.model.6.convtr", 512, 2000), ("decoder.model.7.shortcut.conv", 256, 10000),
          ("decoder.model.7.block.1.conv", 256, 10000), ("decoder.model.7.block.3.conv", 128, 10000), ("decoder.model.9.convtr", 256, 10000),
          ("decoder.model.10.shortcut.conv", 128, 40000), ("decoder.model.10.block.1.conv", 128, 40000), ("decoder.model.10.block.3.conv", 64, 40000),
          ("decoder.model.12.convtr", 128, 40000), ("encoder.model.9.conv", 128, 40000), ("encoder.model.12.conv", 256, 10000), ("encoder.model.15.conv", 512, 2000)]
g = torch.Generator(device="cuda").manual_seed(1)
xs16 = [torch.randn(16, c, t, device="cuda", generator=g) for _, c, t in LAYERS]
xs8 = [(x[:8].contiguous(), x[8:].contiguous()) for x in xs16]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def one():
    for (p, _, _), x in zip(LAYERS, xs16):
        engs[0].layer_forward(p, x, apply_elu=True)

def two():
    for (p, _, _), (xa, xb) in zip(LAYERS, xs8):
        with torch.cuda.stream(streams[0]):
            engs[0].layer_forward(p, xa, apply_elu=True)
        with torch.cuda.stream(streams[1]):
            engs[1].layer_forward(p, xb, apply_elu=True)

for fn, name in ((one, "one stream, 16 utterances"), (two, "two streams, 8 + 8"), (one, "one stream again")):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 200:.3f} ms per pass over {len(LAYERS)} layers")
for e in engs: e.check_status()
