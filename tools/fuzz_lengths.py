"""One-off sweep (not a test): very short and odd utterance lengths through the engine against the CPU oracle (reflect padding of
inputs shorter than the padding, single-frame utterances, lengths around hop multiples)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from helpers import audio, rms, index_report, engine_for, oracle_for, freq_engine_for, freq_oracle_for

bad = 0
cases = [("tiny", 7, False), ("fuzz2", 2, False), ("fuzz10", 10, False), ("tinywn", 9, False), ("tinyss", 5, False), ("fuzz3", 3, False),
         ("tinyfreq", 3, True), ("freqfuzz10", 10, True), ("freqfuzz5", 5, True)]
for cfg, seed, freq in cases:
    m = (freq_engine_for if freq else engine_for)(cfg, seed)
    orc = (freq_oracle_for if freq else oracle_for)(cfg, seed)
    hop = m.engine.hop_length
    lens = sorted(set(list(range(1, 34)) + [hop - 1, hop, hop + 1, 2 * hop - 1, 2 * hop, 2 * hop + 1, 5 * hop + 3, 257, 258, 300, 511, 513]))
    for T in lens:
        wav = audio(2, T, 9000 + T, "noise")
        try:
            o = orc.inference(wav, bit_width=None, use_scale=True)
            oerr = None
        except Exception as ex:
            o, oerr = None, f"{type(ex).__name__}: {str(ex)[:80]}"
        try:
            r = m.inference(wav.cuda().unsqueeze(1), bit_width=None, use_scale=True)
            m.engine.check_status()
            eerr = None
        except Exception as ex:
            r, eerr = None, f"{type(ex).__name__}: {str(ex)[:80]}"
        if oerr or eerr:
            ok = bool(oerr) and bool(eerr)               # the reference path raises -> the engine must refuse too
            bad += not ok
            if not ok or os.environ.get("VERBOSE"):
                print(f"{cfg} T={T}: oracle [{oerr}] engine [{eerr}]{'' if ok else '   <-- CHECK'}", flush=True)
            continue
        rep = index_report(r["code_indices"][0], o["code_indices"][0])
        same_shape = r["recon_speech"].shape == o["recon_speech"].shape
        ref = max(float(o["recon_speech"].double().pow(2).mean().sqrt()), 1e-6) if freq else 1.0
        w = rms(r["recon_speech"], o["recon_speech"]) / ref if same_shape and not rep["frames_bad"] else float("nan")
        flag = "" if same_shape and rep["frames_bad"] <= 1 and not (w > (1e-3 if freq else 1e-4)) else "   <-- CHECK"
        bad += bool(flag)
        if flag:
            print(f"{cfg} T={T}: frames_bad {rep['frames_bad']}/{rep['frames']} wav {w:.2e} shapes {tuple(r['recon_speech'].shape)} {tuple(o['recon_speech'].shape)}{flag}", flush=True)
print("suspicious:", bad)
