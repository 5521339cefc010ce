"""``Text2Audio`` -- same constructor keywords, call signature and return value as the reference's
funcodec/bin/text2audio_inference.py:30-198 (LauraTTS zero-shot generation), running on the MI355X engines:

    from funcodec_amd.bin.text2audio_inference import Text2Audio
    t2a = Text2Audio(config_file="config.yaml", model_file="model.pth", device="cuda", text_emb_model=None, beam_size=1,
                     sampling=25, continual=True, codec_config_file="codec/config.yaml", codec_model_file="codec/model.pth")
    ret_val, decoded_codec = t2a(text, prompt_text, prompt_audio)          # ret_val = {"gen": wav, "gen_only_lm": wav}

plus ``generate_batch`` (up to 16 prompts per engine call: the reference is batch-1 with one host round trip per token).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from pathlib import Path
from typing import Any, List, Optional, Sequence, Union

import numpy as np
import torch

from ..laura import LauraGenMI355X
from ..laura_config import laura_spec_from_config
from .codec_inference import Speech2Token


def build_model_from_file(config_file, model_file, device="cuda", max_positions: int = 2048):
    """Counterpart of Text2AudioGenTask.build_model_from_file (funcodec/tasks/abs_task.py:1895-1947)."""
    import yaml
    with open(config_file, "rt", encoding="utf-8") as f:
        args = yaml.safe_load(f)
    spec = laura_spec_from_config(args)
    model = LauraGenMI355X(spec, device=device, max_positions=max_positions)
    if model_file is not None:
        model.load_state_dict(torch.load(model_file, map_location="cpu"))
    return model, argparse.Namespace(**args)


class Text2Audio:
    """Text2Audio class (drop-in for funcodec.bin.text2audio_inference.Text2Audio)."""

    def __init__(
            self,
            config_file: Union[Path, str] = None,
            model_file: Union[Path, str] = None,
            device: str = "cuda",
            dtype: str = "float32",
            **kwargs
    ):
        if dtype != "float32":
            raise NotImplementedError("the MI355X LauraTTS engine computes in float32 (the reference's default)")
        if device == "cpu":
            raise RuntimeError("funcodec_amd.Text2Audio runs on MI355X only; use the reference for device='cpu'")
        model, model_args = build_model_from_file(config_file, model_file, device, kwargs.get("max_positions", 2048))
        self.model = model
        self.model_args = model_args
        self.device = device
        self.dtype = dtype
        text_emb_model = kwargs.get("text_emb_model")
        self.beam_size = kwargs.get("beam_size", 1)
        self.sampling = kwargs.get("sampling", True)
        self.continual = kwargs.get("continual", True)
        self.tokenize_to_phone = kwargs.get("tokenize_to_phone", False)
        self.exclude_prompt = kwargs.get("exclude_prompt", True)
        self.max_length = kwargs.get("max_length", 30 * 25)          # bin/text2audio_inference.py:168
        if self.tokenize_to_phone:
            raise NotImplementedError("tokenize_to_phone needs the g2p_en package (third party, not in this image); pass phoneme strings")
        if not self.model.vocab_size:
            # embedding-input checkpoints: the reference runs a T5 encoder from `transformers` (:115-135), a third-party model
            # outside this engine; any callable text -> (embeddings [1, L, input_size], lengths [1]) is accepted in its place
            if callable(text_emb_model):
                self.text_emb_model = text_emb_model
            elif text_emb_model:
                self.text_emb_model = self.build_text_emb_model(text_emb_model)
            else:
                raise ValueError("an embedding-input LauraTTS checkpoint needs text_emb_model (a T5 path or a callable)")
        else:
            self.text_emb_model = self.tokenize_text
        codec_kwargs = dict(config_file=kwargs["codec_config_file"], model_file=kwargs["codec_model_file"], device=device)
        self.codec_model = Speech2Token.from_pretrained(model_tag=None, **codec_kwargs)

    # -- text side (bin/text2audio_inference.py:99-135) ------------------------------------------------------------------------
    def token_ids(self, text: str) -> List[int]:
        """tokenize_text (:99-110): whitespace split, tokens missing from the list are dropped."""
        toks = self.model.token_list
        index = getattr(self, "_tok_index", None)
        if index is None:
            index = self._tok_index = {}
            for i, t in enumerate(toks):
                index.setdefault(t, i)              # list.index semantics: first occurrence
        return [index[one] for one in text.strip().split(" ") if one in index]

    def tokenize_text(self, text: str):
        ids = self.token_ids(text)
        logging.info(" ".join(str(x) for x in ids))
        token_idx = torch.tensor(ids, dtype=torch.int64, device=self.model.device)
        text_emb = self.model.token_embedding(token_idx)          # ids: the lookup happens inside the engine (fc_laura_encode)
        return text_emb.unsqueeze(0), torch.tensor([len(ids)], dtype=torch.int64, device=self.model.device)

    def build_text_emb_model(self, model_path: str):
        emb_type = "enc"
        if ":" in model_path:
            model_path, emb_type = model_path.rsplit(":", maxsplit=1)
        from transformers import T5Model, T5Tokenizer
        tokenizer = T5Tokenizer.from_pretrained(model_path)
        model = T5Model.from_pretrained(model_path).to(self.model.device)

        def _forward(text: str):
            inputs = tokenizer(text, return_tensors="pt")
            inputs = {k: v.to(self.model.device) for k, v in inputs.items()}
            with torch.no_grad():
                if emb_type == "enc":
                    outputs = model.encoder(inputs["input_ids"]).last_hidden_state
                else:
                    outputs = model.shared(inputs["input_ids"])
            return outputs, inputs["attention_mask"].sum(dim=1)

        return _forward

    # -- one utterance, exactly the reference's flow (:137-198) ----------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, text: str, prompt_text: str = None, prompt_audio: np.ndarray = None):
        ret, codecs = self.generate_batch([text], None if prompt_text is None else [prompt_text],
                                          None if prompt_audio is None else [prompt_audio])
        return ret[0], codecs[0]

    # -- up to 16 prompts per engine call -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_batch(self, texts: Sequence[str], prompt_texts: Optional[Sequence[str]] = None,
                       prompt_audios: Optional[Sequence[np.ndarray]] = None, seed: Optional[int] = None):
        """Returns (list of {"gen": wav [1, 1, T], "gen_only_lm": wav}, list of decoded_codec [1, T, predict_nq]) like calling the
        reference once per utterance; utterances of one call share the engine passes."""
        m, nq = self.model, self.model.predict_nq
        n = len(texts)
        continual_mode = self.continual and prompt_texts is not None and prompt_audios is not None
        cont, cont_lens = None, None
        if continual_mode:
            texts = [" ".join([p, t]).strip() for p, t in zip(prompt_texts, texts)]
            per = []
            for a in prompt_audios:       # prompt recordings differ in length: one codec call each (no padding inside an utterance)
                a = torch.as_tensor(a, dtype=torch.float32)
                codec = self.codec_model(a, run_mod="encode")[0][0].squeeze(1).transpose(0, 1)      # [T, n_q]
                per.append(codec[:, :nq])
            cont_lens = [int(c.shape[0]) for c in per]
            cont = torch.zeros((n, max(cont_lens), nq), dtype=torch.int64, device=m.device)
            for i, c in enumerate(per):
                cont[i, : c.shape[0]] = c
        # 0. text embeddings, 1. text encoder
        embs, lens = [], []
        for t in texts:
            e, l = self.text_emb_model(t)
            embs.append(e[0])
            lens.append(int(l.reshape(-1)[0]))
        L = max(lens)
        if embs[0].is_floating_point():
            text_in = torch.zeros((n, L, embs[0].shape[-1]), dtype=torch.float32, device=m.device)
        else:
            text_in = torch.full((n, L), -1, dtype=torch.int64, device=m.device)
        for i, e in enumerate(embs):
            text_in[i, : lens[i]] = e[: lens[i]].to(m.device)
        text_outs, _ = m.encode(text_in, torch.tensor(lens))
        # 2. first codec groups, autoregressively
        tokens, out_lens = m.decode_codec_batch(text_outs, lens, self.max_length, self.sampling, cont, cont_lens, seed)
        excl = [(cl if self.exclude_prompt else 0) for cl in (cont_lens or [0] * n)] if continual_mode else [None] * n
        # 3. dense embeddings of all codec groups, then the codec decoder (utterances of equal length share a decoder call)
        emb = m.cal_codec_emb_batch(text_outs, lens, tokens, out_lens)
        rets, codecs = [], []
        for i in range(n):
            dec = tokens[i: i + 1, : out_lens[i]]
            lo = excl[i]
            _, _, gen_only_lm, _ = self.codec_model(dec[:, lo:], bit_width=None, run_mod="decode")
            _, _, gen, _ = self.codec_model(emb[i: i + 1, : out_lens[i]][:, lo:], run_mod="decode_emb")
            rets.append(dict(gen=gen, gen_only_lm=gen_only_lm))
            codecs.append(dec)
        return rets, codecs

    @staticmethod
    def from_pretrained(model_tag: Optional[str] = None, **kwargs: Optional[Any]):
        return Text2Audio(**kwargs)


def save_audio(wav: torch.Tensor, path: Union[Path, str], sample_rate: int, rescale: bool = False):
    """bin/text2audio_inference.py:231-239: like the codec CLI's save_audio with an extra 0.6 gain on the rescaled signal."""
    from ..io import save_audio as _save
    limit = 0.99
    w = torch.as_tensor(wav).detach().float().cpu()
    if rescale:
        mx = w.abs().max()
        w = w * min(limit / mx, 1) * 0.6
    _save(w, str(path), sample_rate, False)


def inference_func(output_dir: Optional[str] = None, batch_size: int = 1, dtype: str = "float32", ngpu: int = 1, seed: int = 0,
                   num_workers: int = 0, log_level: Union[int, str] = "INFO", key_file: Optional[str] = None,
                   config_file: Optional[str] = "config.yaml", model_file: Optional[str] = "model.pth", model_tag: Optional[str] = None,
                   allow_variable_data_keys: bool = True, streaming: bool = False, **kwargs):
    """bin/text2audio_inference.py:242-357: build the model once, return a function over `raw_inputs` = (text,) or
    (text, prompt_text, prompt_audio path | array).  The scp-driven streaming iterator of the reference's data layer is outside
    this engine's scope; `raw_inputs` (its modelscope pipeline form) is supported."""
    if ngpu > 1:
        raise NotImplementedError("only single GPU decoding is supported")
    logging.basicConfig(level=log_level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    torch.manual_seed(seed)
    my_model = Text2Audio.from_pretrained(model_tag=model_tag, config_file=config_file, model_file=model_file, device="cuda", dtype=dtype,
                                          **kwargs)

    def _forward(data_path_and_name_and_type=None, raw_inputs=None, output_dir_v2: Optional[str] = None, param_dict: Optional[dict] = None):
        if raw_inputs is None:
            raise NotImplementedError("pass raw_inputs=(text,) or (text, prompt_text, prompt_audio)")
        inputs = [raw_inputs[0]]
        if len(raw_inputs) == 3:
            audio = raw_inputs[2]
            if isinstance(audio, str):
                from ..io import read_wav, resample
                x, sr = read_wav(audio)
                want = my_model.codec_model.model.quantizer.sampling_rate
                if sr != want:
                    x = resample(torch.from_numpy(x)[None], sr, want)[0].numpy()
                audio = x[np.newaxis, :]
            else:
                audio = np.asarray(audio).squeeze()[None, :]
            inputs += [raw_inputs[1], audio]
        ret_val, _ = my_model(*inputs)
        out_path = output_dir_v2 if output_dir_v2 is not None else output_dir
        if out_path is not None:
            os.makedirs(out_path, exist_ok=True)
            for suffix, wave in ret_val.items():
                save_audio(wave[0], os.path.join(out_path, "utt1_" + suffix + ".wav"), rescale=True,
                           sample_rate=my_model.codec_model.model.quantizer.sampling_rate)
            return []
        return [{"key": "utt1", "value": ret_val}]

    return _forward


def inference(output_dir: Optional[str], batch_size: int = 1, dtype: str = "float32", ngpu: int = 1, seed: int = 0, num_workers: int = 0,
              log_level: Union[int, str] = "INFO", data_path_and_name_and_type=None, key_file: Optional[str] = None,
              config_file: Optional[str] = None, model_file: Optional[str] = None, model_tag: Optional[str] = None,
              allow_variable_data_keys: bool = True, streaming: bool = False, **kwargs):
    """bin/text2audio_inference.py:360-397."""
    pipeline = inference_func(output_dir=output_dir, batch_size=batch_size, dtype=dtype, ngpu=ngpu, seed=seed, num_workers=num_workers,
                              log_level=log_level, key_file=key_file, config_file=config_file, model_file=model_file, model_tag=model_tag,
                              allow_variable_data_keys=allow_variable_data_keys, streaming=streaming,
                              **{k: v for k, v in kwargs.items() if k not in ("raw_inputs", "mode", "gpuid_list")})
    return pipeline(data_path_and_name_and_type, raw_inputs=kwargs.get("raw_inputs", None))


def _int_or_float_or_bool(value: str):
    """funcodec/utils/types.py int_or_float_or_bool: "true"/"false" -> bool, "25" -> int, "0.8" -> float (the --sampling argument)."""
    v = value.strip().lower()
    if v in ("true", "false"):
        return v == "true"
    try:
        return int(v)
    except ValueError:
        return float(v)


def get_parser():
    """Same flags as the reference's CLI (bin/text2audio_inference.py:400-536); `python -m funcodec_amd.bin.text2audio_inference ...`
    accepts the command lines of egs/LibriTTS/text2speech_laura/demo.sh (minus --tokenize_to_phone true: pass phoneme strings)."""
    def str2bool(v):
        return str(v).lower() in ("true", "1", "yes")

    p = argparse.ArgumentParser(description="Text to audio generation", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--log_level", type=lambda x: x.upper(), default="INFO", choices=("CRITICAL", "ERROR", "WARNING", "INFO", "DEBUG", "NOTSET"))
    p.add_argument("--output_dir", type=str, required=False)
    p.add_argument("--ngpu", type=int, default=1)
    p.add_argument("--gpuid_list", type=str, default="0")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--dtype", default="float32", choices=["float16", "float32", "float64"])
    p.add_argument("--num_workers", type=int, default=0)
    g = p.add_argument_group("Input data related")
    g.add_argument("--data_path_and_name_and_type", type=str, required=False, action="append")
    g.add_argument("--raw_inputs", type=str, required=False, action="append")
    g.add_argument("--key_file", type=str, default=None)
    g.add_argument("--allow_variable_data_keys", type=str2bool, default=False)
    g = p.add_argument_group("The model configuration related")
    g.add_argument("--mode", type=str, default="inference mode")
    g.add_argument("--config_file", type=str)
    g.add_argument("--model_file", type=str)
    g.add_argument("--model_tag", type=str)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--beam_size", type=int, default=1)
    g.add_argument("--text_emb_model", type=str, default="./exp/t5-base")
    g.add_argument("--sampling", type=_int_or_float_or_bool, default="true")
    g.add_argument("--codec_config_file", type=str, default=None)
    g.add_argument("--codec_model_file", type=str, default=None)
    g.add_argument("--continual", type=int, default=0)
    g.add_argument("--tokenize_to_phone", type=str2bool, default=False)
    g.add_argument("--exclude_prompt", type=str2bool, default=True)
    return p


def main(cmd=None):
    args = get_parser().parse_args(cmd)
    kwargs = vars(args)
    gpuid = kwargs["gpuid_list"].split(",")[0] or "0"          # one process drives one GPU (bin/text2audio_inference.py:545-556)
    if torch.cuda.is_available():
        torch.cuda.set_device(int(gpuid))
    inference(**kwargs)


if __name__ == "__main__":
    main()
