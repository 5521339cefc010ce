"""End-to-end throughput of the CLI pipeline (encoding_decoding.sh stage 1: wav.scp -> codecs.txt + reconstructed wavs), host I/O
included: N synthetic 10 s wavs on local disk, batch size 16, ds640.  usage: python tools/cli_throughput.py [N] [extra CLI args...]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from funcodec_amd import io as fio
from funcodec_amd.bin.codec_inference import main
from funcodec_amd.synth import make_checkpoint, synthetic_audio

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
extra = sys.argv[2:]
d = tempfile.mkdtemp(prefix="fc_cli_")
cfg_path, pth_path = make_checkpoint(os.path.join(d, "model"), "ds640", 0)
wav = torch.from_numpy(synthetic_audio(16, 160000, 3, "tones"))
scp = os.path.join(d, "wav.scp")
with open(scp, "wt") as f:
    for i in range(N):
        p = os.path.join(d, f"u{i:05d}.wav")
        fio.save_audio(wav[i % 16:i % 16 + 1], p, 16000, rescale=False)
        f.write(f"u{i:05d} {p}\n")
args = ["--ngpu", "1", "--gpuid_list", "0", "--batch_size", "16", "--sampling_rate", "16000", "--config_file", cfg_path,
        "--model_file", pth_path, "--bit_width", "16000", "--need_indices", "true", "--run_mod", "inference",
        "--data_path_and_name_and_type", f"{scp},speech,sound"] + extra
for rep in range(2):                       # first pass warms the page cache / loads the library
    out = os.path.join(d, f"out{rep}.1")
    t0 = time.perf_counter()
    main(["--output_dir", out] + args)
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {N} x 10 s in {dt:.2f} s = {N * 10 / dt:.0f} audio-s/s end to end (engine load included)", flush=True)

# stage 3 of encoding_decoding.sh: decode the codes file back to wavs
codes = os.path.join(d, "out1.1", "codecs.txt")
dargs = ["--ngpu", "1", "--gpuid_list", "0", "--batch_size", "16", "--sampling_rate", "16000", "--config_file", cfg_path,
         "--model_file", pth_path, "--bit_width", "16000", "--run_mod", "decode", "--data_path_and_name_and_type", f"{codes},speech,codec_json"]
for rep in range(2):
    out = os.path.join(d, f"dec{rep}.1")
    t0 = time.perf_counter()
    main(["--output_dir", out] + dargs)
    dt = time.perf_counter() - t0
    print(f"decode pass {rep}: {N} x 10 s in {dt:.2f} s = {N * 10 / dt:.0f} audio-s/s end to end", flush=True)
