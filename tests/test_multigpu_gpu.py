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
