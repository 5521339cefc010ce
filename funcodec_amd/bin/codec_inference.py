"""``Speech2Token`` -- same constructor, call signature and 4-tuple as the reference's
funcodec/bin/codec_inference.py:41-150, running on the MI355X engine.

    from funcodec_amd.bin.codec_inference import Speech2Token
    s2t = Speech2Token("config.yaml", "model.pth", device="cuda")
    code_indices, code_embeddings, recon_speech, sub_quants = s2t(wav)        # run_mod="inference"
"""
from __future__ import annotations

import logging
import math
from pathlib import Path
from typing import Any, Optional, Union

import numpy as np
import torch

from ..config import arch_from_config
from ..model import EncodecMI355X


def build_model_from_file(config_file, model_file, device="cuda"):
    """Counterpart of GANSpeechCodecTask.build_model_from_file (funcodec/tasks/abs_task.py:1895-1947):
    yaml -> architecture, torch.load(model.pth) -> tolerant state_dict load."""
    import argparse
    import yaml
    with open(config_file, "rt", encoding="utf-8") as f:
        args = yaml.safe_load(f)
    arch = arch_from_config(args)
    model = EncodecMI355X(arch, device=device)
    if model_file is not None:
        state = torch.load(model_file, map_location="cpu")
        model.load_state_dict(state)
    return model, argparse.Namespace(**args)


class Speech2Token:
    """Speech2Token class (drop-in for funcodec.bin.codec_inference.Speech2Token)."""

    def __init__(
            self,
            config_file: Union[Path, str] = None,
            model_file: Union[Path, str] = None,
            device: str = "cpu",          # the reference's default (codec_inference.py:53-63); this engine then refuses loudly, see below
            batch_size: int = 1,
            dtype: str = "float32",
            streaming: bool = False,
            sampling_rate: int = 24_000,
            bit_width: int = 24_000,
            check_status: bool = True,
    ):
        if dtype not in ("float16", "float32", "float64"):
            raise ValueError(f"dtype must be float16, float32 or float64, got {dtype!r}")
        # The engine computes in fp32 whatever `dtype` says (index exactness).  The reference converts the MODEL with
        # model.to(dtype) (:78) and feeds the caller's tensors unchanged; here float16 / float64 means: inputs of that dtype are
        # accepted and the floating-point outputs are returned in it (fp32 arithmetic in between).
        # Same signature and default as the reference -- so `Speech2Token(cfg, pth)` does NOT silently pick a GPU: it fails the same way an
        # explicit device="cpu" does, with the fix in the message.
        if str(device).startswith("cpu"):
            raise RuntimeError("funcodec_amd.Speech2Token runs on MI355X only: pass device='cuda' (or 'cuda:<n>'); the default 'cpu' is the "
                               "reference's signature (funcodec/bin/codec_inference.py:56) -- use the reference itself for CPU inference")
        model, model_args = build_model_from_file(config_file, model_file, device)
        self.model = model
        self.model_args = model_args
        self.device = device
        self.dtype = dtype
        self.already_stat_flops = False
        # device-side failures (an out-of-range code index, which the reference's F.embedding raises on; a persistent-LSTM barrier
        # timeout) are recorded by the kernels and can only be read after the stream has drained: by default every call ends with
        # that check, like the reference raises in line.  check_status=False keeps calls asynchronous (the caller then uses
        # model.engine.check_status() itself).
        self.check_status = check_status

    @torch.no_grad()
    def __call__(
            self,
            speech: Union[torch.Tensor, np.ndarray],
            ppg: Optional[Union[torch.Tensor, np.ndarray]] = None,
            need_recon: bool = True,
            bit_width: int = None,
            use_scale: bool = True,
            run_mod: str = "inference",
    ):
        """Returns (code_indices, code_embeddings, recon_speech, sub_quants) like the reference (:86-134)."""
        if ppg is not None:
            raise NotImplementedError("ppg conditioning (codec_semantic_aug) is outside the hot-path scope")
        if isinstance(speech, np.ndarray):
            speech = torch.from_numpy(speech)
        speech = speech.to(self.model.device)
        if run_mod == "inference":
            ret = self.model.inference(speech, need_recon=need_recon, bit_width=bit_width, use_scale=use_scale)
        elif run_mod == "encode":
            ret = self.model.inference_encoding(speech, need_recon=False, bit_width=bit_width)
        elif run_mod == "decode_emb":
            ret = self.model.inference_decoding_emb(speech)
        else:
            q = self.model.quantizer
            bit_per_quant = (q.sampling_rate // q.encoder_hop_length) * int(math.log2(q.codebook_size))
            nq = None
            if bit_width is not None:
                nq = int(max(bit_width // bit_per_quant, 1))
            speech = speech[:, :, :nq]
            logging.info("use %d quantizers.", speech.shape[-1])
            ret = self.model.inference_decoding(speech)
        if self.check_status:
            self.model.engine.check_status(sync=True)
        if self.dtype != "float32":
            td = getattr(torch, self.dtype)
            cast = lambda t: t.to(td) if isinstance(t, torch.Tensor) and t.is_floating_point() else t   # noqa: E731
            ret = dict(ret,
                       code_embeddings=[(cast(q), cast(sc)) for q, sc in ret["code_embeddings"]],
                       recon_speech=cast(ret["recon_speech"]),
                       sub_quants=None if ret["sub_quants"] is None else [cast(x) for x in ret["sub_quants"]])
        return (ret["code_indices"], ret["code_embeddings"], ret["recon_speech"], ret["sub_quants"])

    @staticmethod
    def from_pretrained(model_tag: Optional[str] = None, **kwargs: Optional[Any]):
        """Like the reference (codec_inference.py:136-150: "model_tag ... Currently, not used"): the instance is built from **kwargs
        (config_file / model_file / device ...).  A non-None tag cannot select a model here either -- there is no hub access -- and that is
        said instead of silently dropped."""
        if model_tag is not None:
            logging.warning("Speech2Token.from_pretrained: model_tag=%r is not used (the reference ignores it as well, codec_inference.py:144); "
                            "the model comes from config_file / model_file", model_tag)
            if not (kwargs.get("config_file") and kwargs.get("model_file")):
                raise ValueError(f"Speech2Token.from_pretrained(model_tag={model_tag!r}) needs config_file= and model_file=: model tags are "
                                 "not resolved (no hub access), the tag alone selects nothing")
        return Speech2Token(**kwargs)


class Token2Speech:
    """Convenience wrapper (the reference has no such class; decoding is Speech2Token(run_mod='decode'))."""

    def __init__(self, config_file=None, model_file=None, device: str = "cuda", speech2token: Speech2Token = None):
        self.s2t = speech2token or Speech2Token(config_file, model_file, device=device)

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor, bit_width: int = None) -> torch.Tensor:
        """tokens [B,Tf,n_q] int64 -> waveform [B,1,Tf*hop]"""
        return self.s2t(tokens, bit_width=bit_width, run_mod="decode")[2]


# ================================================================================================
# CLI / batch pipeline: same function names, arguments and output files as the reference's
# funcodec/bin/codec_inference.py:164-580 (egs/LibriTTS/codec/encoding_decoding.sh drives exactly this).
# ================================================================================================
def save_audio(wav, path, sample_rate: int, rescale: bool = False):
    from ..io import save_audio as _save
    _save(wav, str(path), sample_rate, rescale)


def inference_modelscope(
        output_dir: Optional[str] = None,
        batch_size: int = 1,
        dtype: str = "float32",
        ngpu: int = 1,
        seed: int = 0,
        num_workers: int = 0,
        log_level: Union[int, str] = "INFO",
        key_file: Optional[str] = None,
        config_file: Optional[str] = "config.yaml",
        model_file: Optional[str] = "model.pth",
        model_tag: Optional[str] = None,
        allow_variable_data_keys: bool = True,
        streaming: bool = False,
        sampling_rate: int = 16_000,
        bit_width: int = 8_000,
        param_dict: Optional[dict] = None,
        use_scale: Optional[bool] = True,
        **kwargs,
):
    """Returns ``_forward(data_path_and_name_and_type | raw_inputs, output_dir_v2, param_dict)`` (reference :164-382)."""
    import os
    from .. import io as fio
    if param_dict is not None:
        kwargs.update(param_dict)
    if ngpu > 1:
        raise NotImplementedError("only single GPU decoding is supported")       # reference :190-191
    if ngpu < 1:
        raise RuntimeError("funcodec_amd runs on MI355X only (ngpu must be >= 1); use the reference for CPU decoding")
    logging.basicConfig(level=log_level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    torch.manual_seed(seed)
    my_model = Speech2Token.from_pretrained(model_tag=model_tag, config_file=config_file, model_file=model_file,
                                            device="cuda", dtype=dtype, streaming=streaming,
                                            sampling_rate=sampling_rate, bit_width=bit_width,
                                            check_status=False)      # ONE status check per batch, below (not one per call + one per batch)

    def _forward(data_path_and_name_and_type=None, raw_inputs=None, output_dir_v2: Optional[str] = None,
                 param_dict: Optional[dict] = None):
        if param_dict is not None:
            kwargs.update(param_dict)
        if data_path_and_name_and_type is None and raw_inputs is not None:
            uttid = "utt"
            if isinstance(raw_inputs, str):
                uttid = os.path.basename(raw_inputs).rsplit(".")[0]
                raw_inputs, sr = fio.read_wav(raw_inputs)
                if sr != sampling_rate:                                        # librosa.load(sr=...) in the reference (:240)
                    raw_inputs = fio.resample(torch.from_numpy(raw_inputs), sr, sampling_rate).numpy()
            if isinstance(raw_inputs, torch.Tensor):
                raw_inputs = raw_inputs.numpy()
            loader = [([uttid], dict(speech=torch.from_numpy(np.asarray(raw_inputs))[None, :],
                                     speech_lengths=torch.tensor([raw_inputs.shape[0]], dtype=torch.int64)))]
        else:
            loader = fio.iter_batches(data_path_and_name_and_type, batch_size, key_file)
        output_path = output_dir_v2 if output_dir_v2 is not None else output_dir
        if output_path is not None:
            os.makedirs(output_path, exist_ok=True)
        file_sr = kwargs.get("file_sampling_rate")
        should_resample = file_sr not in (None, sampling_rate)               # reference :270-273
        indices_writer, sub_quants_writer = None, None
        ark_indices = kwargs.get("indices_save_type") == "ark"
        if kwargs.get("need_indices"):
            if ark_indices:
                indices_writer = fio.KaldiMatrixWriter(os.path.join(output_path, "indices"))
            else:
                indices_writer = open(os.path.join(output_path, "codecs.txt"), "wt")
        if kwargs.get("need_sub_quants"):
            sub_quants_writer = fio.KaldiMatrixWriter(os.path.join(output_path, "codec_emb"))

        result_list = []
        run_mod = kwargs.get("run_mod", "inference")
        hop = my_model.model.quantizer.encoder_hop_length

        # Host pipeline around the engine (the engine does 160 audio-seconds in 17 ms; reading / padding the next batch and writing
        # the previous one must not sit on its critical path): a loader thread keeps up to 3 batches ready, each finished batch is
        # copied to the host ONCE per output and handed to a small writer pool (wav files: the library's C writer, GIL released);
        # codecs.txt lines and ark entries are appended by this thread in batch order.
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor

        def _prefetch(it, depth=3):
            q: "queue.Queue" = queue.Queue(maxsize=depth)
            end = object()

            def run():
                try:
                    for item in it:
                        q.put(item)
                    q.put(end)
                except BaseException as ex:          # noqa: BLE001 -- re-raised in the consumer
                    q.put(ex)

            threading.Thread(target=run, daemon=True).start()
            while True:
                item = q.get()
                if item is end:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item

        def _write_batch(keys, speech_length, recon_host, tok_host):
            """wav files of one batch + its codecs.txt lines (returned, written in order by the caller)."""
            lines, results = [], []
            for i, key in enumerate(keys):
                if run_mod in ["decode", "decode_emb"]:
                    codec_len = int(speech_length[i])
                    ilen = codec_len * hop
                else:
                    ilen = int(speech_length[i])
                    codec_len = int(math.ceil(ilen / hop))
                recon_wav = recon_host[i][:, :ilen] if recon_host is not None else None
                if output_path is None:
                    results.append({"key": key, "value": recon_wav})
                    continue
                if recon_wav is not None:
                    save_audio(recon_wav, os.path.join(output_path, key + ".wav" if not key.endswith(".wav") else key),
                               rescale=True, sample_rate=file_sr if should_resample else sampling_rate)
                if tok_host is not None and indices_writer is not None and not ark_indices:
                    lines.append(fio.format_codec_line(key, tok_host, i, codec_len))
            return lines, results

        def _drain(job):
            keys, speech_length, tok_host, sq_host, fut = job
            lines, results = fut.result()
            result_list.extend(results)
            if lines:
                indices_writer.write("".join(lines))
            for i, key in enumerate(keys):               # Kaldi ark writers keep file offsets: sequential, in order
                codec_len = int(speech_length[i]) if run_mod in ["decode", "decode_emb"] else int(math.ceil(int(speech_length[i]) / hop))
                if tok_host is not None and indices_writer is not None and ark_indices and output_path is not None:   # [T, n_q] (reference :292-294)
                    mats = [x[:, i, :codec_len].float().numpy().T for x in tok_host]
                    indices_writer(key, np.concatenate(mats, axis=0))
                if sq_host is not None and sub_quants_writer is not None and output_path is not None:                # [T, n_q*D] (reference :301-311)
                    sq = torch.cat(sq_host, dim=-1).permute(1, 3, 0, 2)[i][:codec_len]
                    sub_quants_writer(key, sq.reshape(sq.shape[0], -1).numpy())

        if kwargs.get("stat_flops") and not my_model.already_stat_flops:     # reference :328-342 (thop profile of a 1 s random input)
            eng = my_model.model.engine
            nq = my_model.model.arch.num_quantizers_for_bandwidth(bit_width)
            w = eng.work(1, int(sampling_rate), nq)
            params = sum(int(np.prod(shape)) for shape in eng.expected_tensors().values())
            logging.info(f"Model total MACs: {w['total_flops'] / 2e9:.2f} G (conv {w['conv_flops'] / 2e9:.2f}, lstm {w['lstm_flops'] / 2e9:.2f}, "
                         f"rvq {w['rvq_flops'] / 2e9:.2f}; 1 s of audio, n_q = {nq}), params: {params / 1e6:.2f} M")
            my_model.already_stat_flops = True
        pool = ThreadPoolExecutor(max_workers=4)
        jobs = []
        try:
            for keys, batch in _prefetch(loader):
                if should_resample:                                                 # reference :318-322 (lengths stay in file samples)
                    batch["speech"] = fio.resample(batch["speech"], file_sr, sampling_rate)
                speech_length = batch.pop("speech_lengths")
                bw = param_dict["bit_width"] if param_dict is not None and "bit_width" in param_dict else bit_width
                token_id, token_emb, recon_speech, sub_quants = my_model(**batch, need_recon=True, bit_width=bw,
                                                                         use_scale=use_scale, run_mod=run_mod)
                # device-side failures cannot raise in-line like the reference's F.embedding / asserts do: synchronise and ask the
                # engine before anything of this batch is written (corrupt token files, LSTM barrier timeout)
                my_model.model.engine.check_status(sync=True)
                if should_resample and recon_speech is not None:                    # reference :352-356
                    recon_speech = fio.resample(recon_speech, sampling_rate, file_sr)
                recon_host = recon_speech.cpu() if recon_speech is not None else None
                tok_host = [x.cpu().contiguous() for x in token_id] if (token_id is not None and indices_writer is not None) else None
                sq_host = [x.cpu() for x in sub_quants] if (sub_quants is not None and sub_quants_writer is not None) else None
                jobs.append((keys, speech_length, tok_host, sq_host, pool.submit(_write_batch, keys, speech_length, recon_host, tok_host)))
                while len(jobs) > 3:                                                # bounds the host memory held by queued batches
                    _drain(jobs.pop(0))
            while jobs:
                _drain(jobs.pop(0))
        finally:
            pool.shutdown(wait=True)
        for w in (indices_writer, sub_quants_writer):
            if w is not None:
                w.close()
        return result_list

    return _forward


def inference(output_dir, batch_size, dtype, ngpu, seed, num_workers, log_level, data_path_and_name_and_type, key_file,
              config_file, model_file, model_tag, allow_variable_data_keys=True, streaming=False, sampling_rate=24_000,
              bit_width=24_000, use_scale=True, **kwargs):
    pipeline = inference_modelscope(output_dir=output_dir, batch_size=batch_size, dtype=dtype, ngpu=ngpu, seed=seed,
                                    num_workers=num_workers, log_level=log_level, key_file=key_file, config_file=config_file,
                                    model_file=model_file, model_tag=model_tag,
                                    allow_variable_data_keys=allow_variable_data_keys, streaming=streaming,
                                    sampling_rate=sampling_rate, bit_width=bit_width, use_scale=use_scale, **kwargs)
    return pipeline(data_path_and_name_and_type, raw_inputs=None)


def _str2bool(v: str) -> bool:
    return str(v).lower() in ("true", "1", "yes", "y", "t")


def _str2triple(v: str):
    a, b, c = v.split(",")
    return a.strip(), b.strip(), c.strip()


def get_parser():
    """Same flags (names, types, defaults) as the reference parser (:428-558)."""
    import argparse
    p = argparse.ArgumentParser(description="Speech Tokenizer", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--log_level", type=lambda x: x.upper(), default="INFO",
                   choices=("CRITICAL", "ERROR", "WARNING", "INFO", "DEBUG", "NOTSET"))
    p.add_argument("--output_dir", type=str, required=False)
    p.add_argument("--ngpu", type=int, default=0)
    p.add_argument("--gpuid_list", type=str, default="")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--dtype", default="float32", choices=["float16", "float32", "float64"])
    p.add_argument("--num_workers", type=int, default=0)
    g = p.add_argument_group("Input data related")
    g.add_argument("--data_path_and_name_and_type", type=_str2triple, required=False, action="append")
    g.add_argument("--key_file", type=lambda s: None if s in ("none", "None", "null", "") else s)
    g.add_argument("--allow_variable_data_keys", type=_str2bool, default=False)
    g = p.add_argument_group("The model configuration related")
    g.add_argument("--config_file", type=str)
    g.add_argument("--model_file", type=str)
    g.add_argument("--model_tag", type=str)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--sampling_rate", type=int, default=24_000)
    p.add_argument("--file_sampling_rate", type=int, default=None)
    p.add_argument("--bit_width", type=int, default=16_000)
    p.add_argument("--use_scale", type=_str2bool, default=True)
    g.add_argument("--need_indices", type=_str2bool)
    g.add_argument("--indices_save_type", type=str, default="text")
    g.add_argument("--need_sub_quants", type=_str2bool)
    g.add_argument("--run_mod", type=str, choices=["inference", "encode", "decode", "decode_emb"], default="inference")
    g.add_argument("--stat_flops", type=_str2bool, default=False)
    return p


def main(cmd=None):
    """One process per GPU; the job index and the GPU come from the suffix of --output_dir (``output.JOB``) and
    --gpuid_list exactly as in the reference (:561-580); on ROCm the mask is HIP_VISIBLE_DEVICES and the process then
    uses device 0 (the reference's set_device(int(gpuid)) after masking is only correct for gpuid 0)."""
    import os
    import sys
    print(" ".join(sys.argv), file=sys.stderr)
    args = get_parser().parse_args(cmd)
    if args.file_sampling_rate is None:
        args.file_sampling_rate = args.sampling_rate
    kwargs = vars(args)
    gpus = [g for g in args.gpuid_list.split(",") if g != ""]
    if gpus:
        jobid = 1 if args.output_dir is None else int(args.output_dir.split(".")[-1])
        os.environ["HIP_VISIBLE_DEVICES"] = gpus[(jobid - 1) % len(gpus)]
    kwargs.pop("gpuid_list", None)
    kwargs.pop("stat_flops", None)
    inference(**kwargs)


if __name__ == "__main__":
    main()
