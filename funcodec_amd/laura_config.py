"""LauraTTS (text -> codec tokens -> audio) configuration: the subset of the recipe's ``config.yaml`` the generation path
depends on, validated the way ``funcodec_amd.config`` validates the codec: a key the engine cannot reproduce is refused by
name, never silently ignored.

Replaces, for inference: ``Text2AudioGenTask.build_model`` (funcodec/tasks/text2audio_generation.py:202-247),
``LauraGenModel.__init__`` (funcodec/models/audio_generation/laura_model.py:66-151), ``ConformerEncoder.__init__``
(funcodec/models/encoder/conformer_encoder.py:317-532) and ``TransformerEmbedLM.__init__`` (funcodec/lm/transformer_lm.py:97-181)
for the recipe ``egs/LibriTTS/text2speech_laura/conf/text2audio_codec_lm_nq2_uni_rel_pos.yaml``.
"""
from __future__ import annotations

import copy
import dataclasses
from typing import Any, Dict, List, Optional


@dataclasses.dataclass
class StackSpec:
    """One rel-pos self-attention stack (conformer_encoder.py:317-532 without CNN / macaron modules, or
    TransformerEncoder_s0, transformer_encoder.py:385-654): embed Linear + LayerNorm (+ ReLU) + x*sqrt(d), N pre-norm blocks
    (rel-pos MHA, position-wise FFN), final LayerNorm."""
    idim: int
    d_model: int
    heads: int
    ff: int
    layers: int
    act: str            # "swish" (conformer FFN) | "relu" (TransformerEncoder_s0 FFN)
    embed_relu: bool    # TransformerEncoder_s0's input layer has a ReLU after its LayerNorm (transformer_encoder.py:463-469)
    norm_names: tuple   # state_dict names of (attention norm, FFN norm): ("norm_mha", "norm_ff") | ("norm1", "norm2")


@dataclasses.dataclass
class LauraSpec:
    input_size: int
    vocab_size: int                 # > 0: phoneme / token inputs through `token_embedding`
    codebook_size: int
    codebook_dim: int
    num_quantizers: int             # rows of quantizer_codebook.embed
    predict_nq: int
    pos_emb_type: str               # "split" | "uni" (cal_codec_emb, laura_model.py:312-320)
    bidirectional_inputs: bool
    text_encoder: StackSpec
    codec_lm: StackSpec
    codec_encoder: StackSpec
    token_list: Optional[List[str]] = None

    @property
    def lm_vocab(self) -> int:      # laura_model.py:125
        return (self.codebook_size + 1) * self.predict_nq


def _unsupported(key: str, value: Any, why: str = "") -> NotImplementedError:
    return NotImplementedError(f"LauraTTS config: {key} = {value!r} is not supported by the MI355X engine" + (f" ({why})" if why else ""))


_CONFORMER_DEFAULTS = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=6, input_layer="conv2d",
                           normalize_before=True, concat_after=False, positionwise_layer_type="linear", macaron_style=False,
                           rel_pos_type="legacy", pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn",
                           activation_type="swish", use_cnn_module=True, zero_triu=False, interctc_layer_idx=[],
                           stochastic_depth_rate=0.0)
# dropout rates, kernel sizes of modules that are absent and padding_idx have no effect at inference
_CONFORMER_IGNORED = {"dropout_rate", "positional_dropout_rate", "attention_dropout_rate", "cnn_module_kernel", "padding_idx",
                      "positionwise_conv_kernel_size", "interctc_use_conditioning"}


def _conformer_spec(name: str, kind: Optional[str], conf: Dict[str, Any], idim: int) -> StackSpec:
    if kind != "conformer":
        raise _unsupported(name, kind, "only the recipe's conformer encoder")
    c = dict(_CONFORMER_DEFAULTS)
    for k, v in (conf or {}).items():
        if k in _CONFORMER_IGNORED:
            continue
        if k not in c:
            raise _unsupported(f"{name}_conf.{k}", v, "unknown key")
        c[k] = v
    want = dict(input_layer="linear", normalize_before=True, concat_after=False, positionwise_layer_type="linear",
                macaron_style=False, rel_pos_type="latest", pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn",
                use_cnn_module=False, zero_triu=False)
    for k, v in want.items():
        if c[k] != v:
            raise _unsupported(f"{name}_conf.{k}", c[k], f"the recipe has {v!r}")
    if c["interctc_layer_idx"]:
        raise _unsupported(f"{name}_conf.interctc_layer_idx", c["interctc_layer_idx"])
    if c["activation_type"] not in ("swish", "relu"):
        raise _unsupported(f"{name}_conf.activation_type", c["activation_type"])
    if c["output_size"] % c["attention_heads"]:
        raise ValueError(f"{name}_conf: output_size must be a multiple of attention_heads")
    return StackSpec(idim=idim, d_model=c["output_size"], heads=c["attention_heads"], ff=c["linear_units"], layers=c["num_blocks"],
                     act=c["activation_type"], embed_relu=False, norm_names=("norm_mha", "norm_ff"))


def laura_spec_from_config(cfg: Dict[str, Any]) -> LauraSpec:
    if cfg.get("model", "laura_gen_model") != "laura_gen_model":
        raise _unsupported("model", cfg.get("model"))
    mc = dict(cfg.get("model_conf") or {})
    cc = dict(mc.get("codec_conf") or {})
    K, D, nqs = cc.get("codebook_size", 1024), cc.get("codebook_dim", 128), cc.get("num_quantizers", 32)
    if K != 1024:
        # QuantizerCodebook.codec_index_shift is 1024 * arange(32) whatever the codebook size (laura_model.py:29-30)
        raise _unsupported("model_conf.codec_conf.codebook_size", K, "the reference's index shift is hard-wired to 1024")
    predict_nq = mc.get("predict_nq", 1)
    if not (1 <= predict_nq <= min(nqs, 8)):
        raise _unsupported("model_conf.predict_nq", predict_nq)
    if mc.get("pos_enc", "abs_pos") not in ("abs_pos", "sinusoidal"):
        raise _unsupported("model_conf.pos_enc", mc.get("pos_enc"))
    pos_emb_type = mc.get("pos_emb_type", "split")
    if pos_emb_type not in ("split", "uni"):
        raise _unsupported("model_conf.pos_emb_type", pos_emb_type)
    lm = dict(mc.get("codec_lm_conf") or {})
    if lm.get("name", "transformer") != "transformer":
        raise _unsupported("model_conf.codec_lm_conf.name", lm.get("name"))
    lm_want = dict(pos_enc="rel_pos", selfattention_layer_type="rel_selfattn", pe_type="uni")
    for k, v in lm_want.items():
        if lm.get(k) != v:
            raise _unsupported(f"model_conf.codec_lm_conf.{k}", lm.get(k), f"the recipe has {v!r}")
    for k in ("input_aug_conf", "output_aug_conf"):       # training-time SpecAug only
        lm.pop(k, None)
    if lm.get("input_normalize", False):
        raise _unsupported("model_conf.codec_lm_conf.input_normalize", True)
    if not lm.get("use_decoder", True):
        raise _unsupported("model_conf.codec_lm_conf.use_decoder", False)
    embed_unit = lm.get("embed_unit", 128)
    if embed_unit != D:
        raise _unsupported("model_conf.codec_lm_conf.embed_unit", embed_unit, "must equal codec_conf.codebook_dim")
    lm_vocab = (K + 1) * predict_nq
    if lm.get("text_vocab_size", lm_vocab) < lm_vocab:
        raise _unsupported("model_conf.codec_lm_conf.text_vocab_size", lm.get("text_vocab_size"), "would truncate the codec logits")
    att = lm.get("att_unit", 256)
    heads = lm.get("head", 2)
    if att % heads:
        raise ValueError("codec_lm_conf: att_unit must be a multiple of head")
    lm_spec = StackSpec(idim=D, d_model=att, heads=heads, ff=lm.get("unit", 1024), layers=lm.get("layer", 4), act="relu",
                        embed_relu=True, norm_names=("norm1", "norm2"))
    input_size = int(cfg.get("input_size") or 0)
    token_list = cfg.get("token_list")
    if isinstance(token_list, str):
        raise _unsupported("token_list", token_list, "give the list itself, as the released config.yaml does")
    vocab = len(token_list) if token_list else 0
    if cfg.get("text_encoder", "transformer") is None:
        raise _unsupported("text_encoder", None, "the recipe encodes text with a conformer")
    te = _conformer_spec("text_encoder", cfg.get("text_encoder", "transformer"), cfg.get("text_encoder_conf"), input_size)
    ce = _conformer_spec("codec_encoder", cfg.get("codec_encoder", "transformer"), cfg.get("codec_encoder_conf"), D)
    for s, n in ((te, "text_encoder"), (lm_spec, "codec_lm"), (ce, "codec_encoder")):
        dk = s.d_model // s.heads
        if dk not in (32, 64):
            raise _unsupported(f"{n}: head dimension", dk, "the attention kernels are built for 32 and 64")
        if s.d_model % 64 or s.ff % 64 or s.idim % 8:
            raise _unsupported(f"{n}: sizes", (s.idim, s.d_model, s.ff), "d_model and linear_units must be multiples of 64, the input size of 8")
        if s.d_model > 1024:
            raise _unsupported(f"{n}: d_model", s.d_model, "the LayerNorm kernels and the sampler's input layer hold rows of at most 1024 channels")
    if lm_spec.ff > 2112:
        raise _unsupported("codec_lm: unit", lm_spec.ff, "the decoding step's GEMV stages 17 rows of ff floats in LDS: at most 2112")
    return LauraSpec(input_size=input_size, vocab_size=vocab, codebook_size=K, codebook_dim=D, num_quantizers=nqs,
                     predict_nq=predict_nq, pos_emb_type=pos_emb_type, bidirectional_inputs=bool(lm.get("bidirectional_inputs", False)),
                     text_encoder=te, codec_lm=lm_spec, codec_encoder=ce, token_list=list(token_list) if token_list else None)


def laura_recipe_config(name: str) -> Dict[str, Any]:
    """``laura``: egs/LibriTTS/text2speech_laura/conf/text2audio_codec_lm_nq2_uni_rel_pos.yaml (84 M parameters; T5 embeddings of
    width 1536 as text input).  ``lauraphn``: the same nets over a phoneme token list (``vocab_size`` > 0, input_size 256).
    ``tinylaura`` / ``tinylauraphn``: the same structure, small (fast parity cases)."""
    conf_big = dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                    positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="linear", normalize_before=True,
                    rel_pos_type="latest", pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn", use_cnn_module=False)
    cfg = {
        "input_size": 1536, "use_preprocessor": True, "audio_max_duration": 60, "codec_token_rate": 25,
        "init": None, "token_list": None,      # as a saved config.yaml carries them (Text2AudioGenTask.build_model reads both)
        "text_encoder": "conformer", "text_encoder_conf": copy.deepcopy(conf_big),
        "codec_encoder": "conformer", "codec_encoder_conf": copy.deepcopy(conf_big),
        "model": "laura_gen_model",
        "model_conf": {
            "codec_sampling_ratio": 0.5, "lsm_weight": 0.0, "length_normalized_loss": True, "predict_nq": 2,
            "codec_conf": {"num_quantizers": 32, "codebook_size": 1024, "codebook_dim": 128},
            "codec_lm_conf": {"name": "transformer", "pos_enc": "rel_pos", "selfattention_layer_type": "rel_selfattn",
                              "embed_unit": 128, "att_unit": 512, "head": 8, "unit": 2048, "layer": 12, "dropout_rate": 0.1,
                              "pe_type": "uni", "bidirectional_inputs": True, "codec_groups": 1},
        },
    }
    if name == "laura":
        return cfg
    if name == "lauraphn":
        cfg["input_size"] = 256
        cfg["token_list"] = phoneme_token_list()
        return cfg
    if name in ("tinylaura", "tinylauraphn", "tinylaurauni"):
        small = dict(output_size=128, attention_heads=2, linear_units=256, num_blocks=2)
        cfg["text_encoder_conf"].update(small)
        cfg["codec_encoder_conf"].update(small)
        cfg["codec_encoder_conf"]["attention_heads"] = 4          # head dimension 32
        cfg["model_conf"]["codec_conf"]["num_quantizers"] = 4
        cfg["model_conf"]["codec_lm_conf"].update(att_unit=128, head=2, unit=256, layer=3)
        cfg["input_size"] = 24
        if name == "tinylauraphn":
            cfg["input_size"] = 32
            cfg["token_list"] = phoneme_token_list()
        if name == "tinylaurauni":
            cfg["model_conf"]["pos_emb_type"] = "uni"
            cfg["model_conf"]["predict_nq"] = 1
            cfg["model_conf"]["codec_lm_conf"]["bidirectional_inputs"] = False
        return cfg
    raise KeyError(name)


def phoneme_token_list() -> List[str]:
    """A phoneme inventory in the style of the recipe's ``token_list`` (ARPAbet with stress marks + punctuation); the released
    list is not in the reference tree, so this one only fixes a vocabulary size and spelling for the synthetic checkpoints."""
    vowels = ["AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW"]
    cons = ["B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T", "TH", "V", "W", "Y", "Z", "ZH"]
    toks = ["<blank>", "<unk>"] + [v + s for v in vowels for s in ("0", "1", "2")] + cons + [",", ".", "?", "!", "<space>"]
    return toks
