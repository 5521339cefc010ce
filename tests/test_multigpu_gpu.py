"""Hardware multi-GPU path: 2 ranks over NCCL (= RCCL on ROCm) when the box has >= 2 GPUs (skipped on the 1-GPU test boxes); on ANY box a
ONE-rank RCCL communicator runs the N-rank code path (init with a bound device, barrier, the padded all_gather_into_tensor of the codes, a
concurrent RCCL kernel next to the persistent LSTM) so that the first multi-GPU run does not also debut the library calls.
The gathered codes must equal the single-GPU result bit for bit, and the persistent LSTM kernel must survive a concurrent RCCL
kernel on another stream (its grid barrier needs every workgroup resident)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the node")
def test_two_ranks_over_rccl_match_single_gpu(tmp_path):
    out = str(tmp_path / "result.txt")
    port = 29600 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), os.path.join(ROOT, "tests", "mgpu_worker.py"), out],
                   check=True, cwd=ROOT, env=env, timeout=600)
    assert open(out).read().startswith("ok ranks_seen=2")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the node")
def test_bench_gpus_2_self_launches_and_reports_two_ranks():
    """`python bench.py --gpus 2` as the driver runs it for the scaling curve: self-launch, 2 ranks over RCCL, ONE JSON line whose
    `config.ranks_seen` is 2 and which carries the per-rank / gather-only diagnostics."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FC_BENCH_UTTS="32")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-event-profile"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and out["config"]["global_utterances"] == 64
    mg = out["multi_gpu"]
    assert len(mg["rank_ms_per_step"]) == 2 and mg["gather_ms"] > 0 and mg["backend"] == "nccl" and mg["rccl_version"]


def test_one_rank_rccl_communicator_runs_the_n_rank_gather_path(tmp_path):
    """torchrun with ONE rank on the box's GPU: `init_process_group("nccl", device_id=...)`, barrier, all_reduce on a side stream while the
    engine's persistent LSTM runs, and `parallel._gather_ranks` (the body of `gather_codes` for N > 1: pad, ONE all_gather_into_tensor,
    re-assemble) with and without `shard_sizes` -- its result must be the input.  What cannot be rehearsed here is xGMI itself."""
    out = str(tmp_path / "one_rank.txt")
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, '.')\n"
        "rank, local, world = int(os.environ['RANK']), int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert world == 1\n"
        "torch.cuda.set_device(local)\n"
        "dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))\n"
        "from funcodec_amd.config import arch_from_config, recipe_config\n"
        "from funcodec_amd.model import EncodecMI355X\n"
        "from funcodec_amd.parallel import _gather_ranks, gather_codes\n"
        "from funcodec_amd.synth import make_state_dict, synthetic_audio\n"
        "arch = arch_from_config(recipe_config('ds320'))\n"
        "m = EncodecMI355X(arch, 'cuda:0'); m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(arch, 0).items()})\n"
        "wav = torch.from_numpy(synthetic_audio(3, 32000, 77, 'tones')).cuda()\n"
        "side = torch.cuda.Stream(); junk = torch.ones(1 << 22, device='cuda')\n"
        "for it in range(3):\n"
        "    with torch.cuda.stream(side):\n"
        "        for _ in range(4): dist.all_reduce(junk)\n"
        "    r = m.engine.encode_decode(wav, 32)\n"
        "    g1 = _gather_ranks(r['codes'], dist, shard_sizes=[3])\n"
        "    g2 = _gather_ranks(r['codes'], dist)\n"
        "dist.barrier(); torch.cuda.synchronize(); m.engine.check_status()\n"
        "assert torch.equal(g1, r['codes']) and torch.equal(g2, r['codes']) and torch.equal(gather_codes(r['codes'], dist), r['codes'])\n"
        "t = torch.tensor([1.5], dtype=torch.float64, device='cuda'); lst = [torch.zeros_like(t)]; dist.all_gather(lst, t)\n"
        "assert float(lst[0].item()) == 1.5\n"
        "open(sys.argv[1], 'w').write('ok backend=%s rccl=%s' % (dist.get_backend(), '.'.join(map(str, torch.cuda.nccl.version()))))\n"
        "dist.destroy_process_group()\n")
    port = 29300 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    script = str(tmp_path / "one_rank.py")
    open(script, "w").write(code)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), script, out], check=True, cwd=ROOT, env=env, timeout=600)
    assert open(out).read().startswith("ok backend=nccl")
