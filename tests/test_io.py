"""CPU tests of the batch I/O / wire formats around the hot path."""
import json
import os

import numpy as np
import pytest
import torch

from funcodec_amd import io as fio


def test_wav_round_trip_and_save_audio_rescale(tmp_path):
    rng = np.random.default_rng(0)
    x = (0.3 * rng.standard_normal(1000)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    fio.save_audio(torch.from_numpy(x)[None], p, 16000, rescale=True)
    y, sr = fio.read_wav(p)
    assert sr == 16000 and y.shape == x.shape
    scale = min(0.99 / np.abs(x).max(), 1.0)                       # codec_inference.py:153-161
    assert np.abs(y - x * scale).max() <= 1.0 / 32768 + 1e-7
    # without rescale: clamp to +-0.99
    fio.save_audio(torch.tensor([[2.0, -2.0, 0.5]]), p, 16000, rescale=False)
    y, _ = fio.read_wav(p)
    assert np.allclose(y, [0.99, -0.99, 0.5], atol=1.0 / 32768)


def test_kaldi_ark_scp_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    mats = {f"utt{i}": rng.standard_normal((3 + i, 5)).astype(np.float32) for i in range(4)}
    with fio.KaldiMatrixWriter(str(tmp_path / "indices")) as w:
        for k, m in mats.items():
            w(k, m)
    scp = fio.read_scp(str(tmp_path / "indices.scp"))
    assert [k for k, _ in scp] == list(mats)
    for k, spec in scp:
        assert np.array_equal(fio.load_kaldi_mat(spec), mats[k])
    raw = open(tmp_path / "indices.ark", "rb").read()
    assert raw.startswith(b"utt0 \0BFM \x04")                       # kaldiio binary float-matrix layout


def test_codec_jsonl_format_and_inverse():
    codes = torch.arange(2 * 3 * 7).reshape(2, 3, 7)                 # [n_q, B, T]
    line = fio.format_codec_line("k1", [codes], batch_id=1, length=5)
    key, payload = line.rstrip("\n").split(" ", 1)
    arr = json.loads(payload)
    assert key == "k1" and np.array(arr).shape == (1, 2, 5)          # n_frame x n_q x T
    back = fio.load_codec_json(payload)                              # [T, n_q] like iterable_dataset.py:54-58
    assert np.array_equal(back, codes[:, 1, :5].numpy().T)


def test_wrap_padding_matches_numpy_wrap(tmp_path):
    rng = np.random.default_rng(2)
    lens = [700, 1000, 333]
    scp = tmp_path / "wav.scp"
    with open(scp, "wt") as f:
        for i, n in enumerate(lens):
            p = str(tmp_path / f"u{i}.wav")
            fio.save_audio(torch.from_numpy((0.2 * rng.standard_normal(n)).astype(np.float32))[None], p, 16000, rescale=False)
            f.write(f"u{i} {p}\n")
    batches = list(fio.iter_batches([(str(scp), "speech", "sound")], batch_size=2))
    assert [k for k, _ in batches] == [["u0", "u1"], ["u2"]]
    keys, b = batches[0]
    assert b["speech"].shape == (2, 1000) and b["speech_lengths"].tolist() == [700, 1000]
    x0, _ = fio.read_wav(str(tmp_path / "u0.wav"))
    assert np.array_equal(b["speech"][0].numpy(), np.pad(x0, (0, 300), mode="wrap"))   # nets_utils.py:65-98
    assert np.array_equal(b["speech"][0, 700:1000].numpy(), x0[:300])


def test_cli_parser_has_the_reference_flags():
    from funcodec_amd.bin.codec_inference import get_parser
    want = {"--log_level", "--output_dir", "--ngpu", "--gpuid_list", "--seed", "--dtype", "--num_workers",
            "--data_path_and_name_and_type", "--key_file", "--allow_variable_data_keys", "--config_file", "--model_file",
            "--model_tag", "--batch_size", "--sampling_rate", "--file_sampling_rate", "--bit_width", "--use_scale",
            "--need_indices", "--indices_save_type", "--need_sub_quants", "--run_mod", "--stat_flops"}   # reference :428-558
    have = {s for a in get_parser()._actions for s in a.option_strings}
    assert want <= have
    ns = get_parser().parse_args(["--data_path_and_name_and_type", "a.scp,speech,sound", "--need_indices", "true"])
    assert ns.data_path_and_name_and_type == [("a.scp", "speech", "sound")] and ns.need_indices is True
    assert ns.bit_width == 16000 and ns.sampling_rate == 24000 and ns.run_mod == "inference"


def test_resample_matches_the_published_sinc_hann_algorithm_properties():
    """fio.resample restates torchaudio.functional.resample (sinc_interp_hann, width 6, rolloff 0.99); torchaudio is not
    installed here, so the checks are the algorithm's properties: identity, output length ceil(new*T/orig), a band-limited
    tone survives down- and up-sampling, agreement with scipy's polyphase resampler, batch dimensions kept."""
    import math
    from scipy.signal import resample_poly
    x = torch.randn(3, 1001)
    assert fio.resample(x, 16000, 16000) is x
    for o, n in ((24000, 16000), (16000, 24000), (44100, 16000), (8000, 16000)):
        y = fio.resample(x, o, n)
        assert y.shape == (3, math.ceil(n * 1001 / o)) and torch.isfinite(y).all()
    assert fio.resample(torch.randn(2, 1, 500), 16000, 8000).shape == (2, 1, 250)
    t = torch.arange(24000, dtype=torch.float64) / 24000
    tone = torch.sin(2 * math.pi * 440 * t).float()
    down = fio.resample(tone, 24000, 16000)
    ideal = torch.sin(2 * math.pi * 440 * torch.arange(16000, dtype=torch.float64) / 16000).float()
    assert (down[100:-100] - ideal[100:-100]).abs().max() < 1e-3
    assert np.abs(down.numpy()[200:-200] - resample_poly(tone.double().numpy(), 2, 3)[200:-200]).max() < 2e-3
    up = fio.resample(down, 16000, 24000)
    assert (up[300:-300] - tone[300:up.numel() - 300]).abs().max() < 2e-3
    # a tone above the new Nyquist is removed (anti-aliasing), not folded back
    hi = torch.sin(2 * math.pi * 11000 * t).float()
    assert fio.resample(hi, 24000, 16000)[200:-200].abs().max() < 2e-2


def test_native_wire_format_writers_are_byte_identical(tmp_path):
    """fc_format_codec_json == json.dumps(x.tolist()) and fc_write_wav_pcm16 == the torch formula of save_audio
    (codec_inference.py:153-161, 295-299), byte for byte."""
    import json
    import wave
    from funcodec_amd import io as fio
    assert fio._native() is not None
    g = torch.Generator().manual_seed(0)
    tok = [torch.randint(0, 1024, (8, 3, 57), generator=g)]
    tok[0][0, 1, 0] = 0
    tok[0][7, 2, 56] = 1023
    for b, n in ((0, 57), (1, 1), (2, 56), (1, 0)):
        assert fio.format_codec_line("utt", tok, b, n) == "utt " + json.dumps([tok[0][:, b, :n].numpy().tolist()]) + "\n"

    def torch_save(wav, path, sr, rescale):
        limit = 0.99
        mx = wav.abs().max()
        w = (wav * min(limit / mx, 1) if mx > 0 else wav) if rescale else wav.clamp(-limit, limit)
        pcm = torch.clamp((w * 32768.0).round(), -32768, 32767).to(torch.int16).numpy()
        with wave.open(path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr); f.writeframes(np.ascontiguousarray(pcm.T).tobytes())

    # many peaks with rescale: `limit / mx` on a tensor is reciprocal-then-multiply, 1 ulp away from a plain division for ~25 % of them
    for i in range(60):
        x = torch.randn(1, 2000 + 37 * i, generator=g) * (0.3 + 0.11 * i)
        a, b = str(tmp_path / "a.wav"), str(tmp_path / "b.wav")
        fio.save_audio(x, a, 16000, rescale=True)
        torch_save(x, b, 16000, True)
        assert open(a, "rb").read() == open(b, "rb").read(), i
    for rescale in (True, False):
        for amp in (0.3, 2.5, 0.0, 1e-6):
            x = torch.randn(1, 4097, generator=g) * amp
            x[0, 5] = 0.5 / 32768.0 * (1 if amp else 0)            # a round-half-to-even sample
            a, b = str(tmp_path / "a.wav"), str(tmp_path / "b.wav")
            fio.save_audio(x, a, 16000, rescale=rescale)
            torch_save(x, b, 16000, rescale)
            assert open(a, "rb").read() == open(b, "rb").read(), (rescale, amp)
            y, sr = fio.read_wav(a)                                 # the stdlib fast path reads it back
            assert sr == 16000 and y.shape == (4097,)
