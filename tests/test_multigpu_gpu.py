"""Hardware multi-GPU path: 2 ranks over NCCL (= RCCL on ROCm) when the box has >= 2 GPUs (skipped on the 1-GPU test boxes).
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
