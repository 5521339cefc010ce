"""VERDICT r5 #6b: how the CPU baseline that travels to the GPU box (`cpu_baseline.kind = "port"`: oracle/torch_oracle.py, the restatement
over the same ATen CPU kernels) compares with the reference's OWN `Speech2Token(device="cpu")` on the same batch, same thread count, same box.
The reference cannot travel (it only exists in the build container), so the ratio is measured HERE once and carried in bench.py's
`cpu_baseline.port_over_reference` (profiles/r06_port_over_reference.json).

  python tools/port_over_reference.py [utterances=4] [threads=8] > profiles/r06_port_over_reference.json
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402,F401  (stubs the packages the reference imports and this image lacks)

ref_shim.install()
import torch  # noqa: E402

from funcodec_amd.synth import synthetic_audio  # noqa: E402


def main():
    utts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
    torch.set_num_threads(threads)
    import make_golden  # noqa: E402  (build_reference: the real Speech2Token over the synthetic seeded checkpoint)
    from torch_oracle import Oracle
    samples = 160000
    wav = torch.from_numpy(synthetic_audio(utts, samples, 1234))
    with tempfile.TemporaryDirectory() as tmp:
        s2t, cfg, sd = make_golden.build_reference("ds640", 0, 1.0, tmp)
        orc = Oracle(cfg, {k: torch.from_numpy(v) for k, v in sd.items()})

        def run_ref():
            with torch.no_grad():
                return s2t(wav, need_recon=True, bit_width=None, use_scale=True, run_mod="inference")

        def run_port():
            return orc.inference(wav, bit_width=None, use_scale=True)

        res = {}
        for name, fn in (("reference", run_ref), ("port", run_port), ("reference_again", run_ref), ("port_again", run_port)):
            fn()                                  # warm-up
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = fn()
                ts.append(time.perf_counter() - t0)
            res[name] = sorted(ts)[1]
        # same answers (the oracle is pinned bit-for-bit by oracle/make_golden.py; asserted again on this batch)
        r = run_ref()
        o = run_port()
        same = bool(torch.equal(r[0][0], o["code_indices"][0]))
    ref_s = min(res["reference"], res["reference_again"])
    port_s = min(res["port"], res["port_again"])
    audio_s = utts * samples / 16000.0
    print(json.dumps({
        "what": "reference Speech2Token(device='cpu') vs oracle/torch_oracle.py (bench.py's cpu_baseline 'port'), same batch / threads / box "
                "(build container), ds640 synthetic seeded checkpoint, run_mod=inference, n_q=32, median of 3 after a warm-up, best of two rounds",
        "utterances": utts, "seconds_of_audio": audio_s, "threads": threads, "torch": torch.__version__,
        "reference_s": round(ref_s, 3), "port_s": round(port_s, 3),
        "reference_audio_s_per_s": round(audio_s / ref_s, 2), "port_audio_s_per_s": round(audio_s / port_s, 2),
        "port_over_reference": round((audio_s / port_s) / (audio_s / ref_s), 3),
        "indices_identical": same,
        "raw": {k: round(v, 3) for k, v in res.items()}}))


if __name__ == "__main__":
    main()
