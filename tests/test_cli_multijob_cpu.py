"""The CLI's own multi-GPU story, rehearsed on CPU: the reference recipe's encoding stage
(egs/LibriTTS/codec/encoding_decoding.sh:59-101) splits wav.scp into `inference_nj` key files and launches one
`python -m funcodec.bin.codec_inference` process per job with `--gpuid_list`, `--key_file keys.JOB.scp` and
`--output_dir logdir/output.JOB`; the job index and the GPU come from the SUFFIX of --output_dir
(bin/codec_inference.py:569-579), and the per-job codecs.txt files are concatenated afterwards.

Two real processes run `funcodec_amd.bin.codec_inference.main()` with exactly those arguments.  There is no GPU here, so the
processes substitute a stand-in for `Speech2Token.from_pretrained` (a per-utterance function of the samples, so the result of a
job does not depend on which utterances share its batches); everything else -- argument parsing, job / GPU selection, the scp
reader, wrap-pad collate, the loader thread, the writer pool, codecs.txt -- is the product code.  What is checked: each job
masks the GPU the reference would give it, writes only its own keys, and `cat output.*/codecs.txt` equals a one-job run."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, math, os, sys
import torch
sys.path.insert(0, sys.argv[1])
import funcodec_amd.bin.codec_inference as ci

HOP, NQ = 320, 4

class _Q:
    encoder_hop_length = HOP
    sampling_rate = 16000
    codebook_size = 1024

class _Eng:
    def check_status(self, sync=False):
        return None

class _Model:
    quantizer = _Q()
    engine = _Eng()

class StandIn:
    """Speech2Token's call contract (bin/codec_inference.py:86-134) with codes that are a function of the frame's own samples."""
    def __init__(self):
        self.model = _Model()
        self.already_stat_flops = False
    def __call__(self, speech, need_recon=True, bit_width=None, use_scale=True, run_mod="inference", **kw):
        B, T = speech.shape
        Tf = int(math.ceil(T / HOP))
        x = torch.nn.functional.pad(speech, (0, Tf * HOP - T)).reshape(B, Tf, HOP)
        base = (x.abs().sum(-1) * 1000.0).floor().long()
        codes = torch.stack([(base + 17 * q) % 1024 for q in range(NQ)], 0)          # [n_q, B, Tf]
        return [codes], None, None, None

ci.Speech2Token.from_pretrained = staticmethod(lambda **kw: StandIn())
ci.main(sys.argv[3:])
with open(sys.argv[2], "w") as f:
    json.dump({"HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES")}, f)
'''


def _write_wavs(d, n):
    from funcodec_amd import io as fio
    rng = np.random.default_rng(7)
    lines = []
    for i in range(n):
        x = (0.1 * rng.standard_normal(320 * (9 + 2 * i))).astype(np.float32)     # whole frames: no frame sees the wrap-padding of its batch
        p = os.path.join(d, f"utt{i:02d}.wav")
        fio.save_audio(x[None], p, 16000, False)
        lines.append(f"utt{i:02d} {p}\n")
    return lines


def _job(tmp, job, key_file, wav_scp, gpuid_list, env_out):
    argv = [sys.executable, "-c", STUB, ROOT, env_out,
            "--batch_size", "3", "--ngpu", "1", "--gpuid_list", gpuid_list,
            "--data_path_and_name_and_type", f"{wav_scp},speech,sound", "--key_file", key_file,
            "--config_file", "unused.yaml", "--model_file", "unused.pth",
            "--output_dir", os.path.join(tmp, "logdir", f"output.{job}"),
            "--sampling_rate", "16000", "--file_sampling_rate", "16000", "--bit_width", "16000",
            "--need_indices", "true", "--need_sub_quants", "false", "--use_scale", "false",
            "--indices_save_type", "text", "--run_mod", "encode"]
    env = {k: v for k, v in os.environ.items() if k != "HIP_VISIBLE_DEVICES"}
    return subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_two_cli_jobs_split_the_scp_and_concatenate_to_the_one_job_result(tmp_path):
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, "logdir"))
    lines = _write_wavs(tmp, 7)
    wav_scp = os.path.join(tmp, "wav.scp")
    open(wav_scp, "w").writelines(lines)
    # utils/split_scp.pl: contiguous pieces, the first `remainder` pieces one line longer
    splits = [lines[:4], lines[4:]]
    for j, part in enumerate(splits, 1):
        open(os.path.join(tmp, "logdir", f"keys.{j}.scp"), "w").writelines(part)
    procs = [_job(tmp, j, os.path.join(tmp, "logdir", f"keys.{j}.scp"), wav_scp, "3,5", os.path.join(tmp, f"env.{j}.json")) for j in (1, 2)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
    # job JOB takes gpuid_list[(JOB - 1) % len]: the reference's rule (:573-576), as HIP_VISIBLE_DEVICES on ROCm
    assert json.load(open(os.path.join(tmp, "env.1.json")))["HIP_VISIBLE_DEVICES"] == "3"
    assert json.load(open(os.path.join(tmp, "env.2.json")))["HIP_VISIBLE_DEVICES"] == "5"
    per_job = [open(os.path.join(tmp, "logdir", f"output.{j}", "codecs.txt")).read() for j in (1, 2)]
    for j, part in enumerate(splits):
        keys = [ln.split()[0] for ln in part]
        assert [ln.split(" ", 1)[0] for ln in per_job[j].splitlines()] == keys         # only its own keys, in scp order
    # the recipe's `cat output.*/codecs.txt`, against ONE job over the whole list (other batch composition: 3 + 3 + 1 instead of 3 + 1 | 3)
    one = _job(tmp, 9, wav_scp, wav_scp, "0", os.path.join(tmp, "env.9.json"))
    out, err = one.communicate(timeout=300)
    assert one.returncode == 0, err[-3000:]
    whole = open(os.path.join(tmp, "logdir", "output.9", "codecs.txt")).read()
    assert "".join(per_job) == whole
    # wire format: "<uttid> [[[...T ints...] x n_q]]" (write_indices :288-299), frames = ceil(len / hop)
    key, payload = whole.splitlines()[0].split(" ", 1)
    arr = np.array(json.loads(payload))
    assert key == "utt00" and arr.shape == (1, 4, 9)
