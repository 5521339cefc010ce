"""Tuning aid: time of the fused RVQ kernel against the number of rows (16 rows per workgroup): does a second workgroup per CU overlap
the per-stage argmax / gather tail of the first?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funcodec_amd.config import arch_from_config, recipe_config
from funcodec_amd.model import EncodecMI355X
from funcodec_amd.synth import make_state_dict

arch = arch_from_config(recipe_config("ds640"))
m = EncodecMI355X(arch, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(arch, 0).items()})
g = torch.Generator().manual_seed(1)
for n in (2000, 4000, 4096, 8000, 8192, 16000, 16384, 32768):
    x = torch.randn(n, 128, generator=g).cuda()
    for _ in range(3):
        m.engine.rvq_encode(x, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.engine.rvq_encode(x, 32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"rows {n:6d}  workgroups {n // 16:5d}  {dt * 1e6:8.1f} us  {dt * 1e6 / (n / 16 / 256):7.1f} us per (workgroup per CU)")
