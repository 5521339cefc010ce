"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (only tests/, bench.py's cpu_baseline leg,
__graft_entry__.smoke() and oracle/make_golden_laura.py use it).

CPU restatement of the reference's LauraTTS generation path over the same ATen CPU kernels the reference calls, function by
function, each citing the reference file:line it follows.  Pinned against the REAL reference (LauraGenModel built by
Text2AudioGenTask.build_model from the recipe yaml, imported through oracle/ref_shim.py) by oracle/make_golden_laura.py, which
asserts `torch.equal` on every deterministic output while it writes tests/golden/laura_*.npz.

Path (funcodec/bin/text2audio_inference.py:137-198): text embedding -> LauraGenModel.encode -> decode_codec (autoregressive,
the reference re-scores the whole prefix per token without a KV cache) -> syn_audio (cal_codec_emb: NAR conformer) ->
Speech2Token decode_emb.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x


def sinusoid_table(n_pos: int, d_model: int, sign: float = 1.0) -> torch.Tensor:
    """pe[pos, 0::2] = sin(sign*pos*div), pe[pos, 1::2] = cos(sign*pos*div) in fp32, exactly as
    funcodec/modules/embedding.py:63-77 (PositionalEncoding.extend_pe) / :293-306 (RelPositionalEncoding.extend_pe)."""
    pe = torch.zeros(n_pos, d_model)
    position = torch.arange(0, n_pos, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    if sign > 0:
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
    else:
        pe[:, 0::2] = torch.sin(-1 * position * div_term)
        pe[:, 1::2] = torch.cos(-1 * position * div_term)
    return pe


def rel_pos_emb(T: int, d_model: int) -> torch.Tensor:
    """RelPositionalEncoding.forward's second output (embedding.py:309-331): [1, 2T-1, d]; row idx <-> relative position
    T-1-idx (keys to the left of the query are positive)."""
    pos = torch.flip(sinusoid_table(T, d_model, +1.0), [0])
    neg = sinusoid_table(T, d_model, -1.0)[1:]
    return torch.cat([pos, neg], dim=0).unsqueeze(0)


def rel_shift(x: torch.Tensor) -> torch.Tensor:
    """RelPositionMultiHeadedAttention.rel_shift (funcodec/modules/attention.py:240-263), zero_triu False."""
    zero_pad = torch.zeros((*x.size()[:3], 1), dtype=x.dtype)
    x_padded = torch.cat([zero_pad, x], dim=-1)
    x_padded = x_padded.view(*x.size()[:2], x.size(3) + 1, x.size(2))
    return x_padded[:, :, 1:].view_as(x)[:, :, :, : x.size(-1) // 2 + 1]


def make_pad_mask(lengths, maxlen=None) -> torch.Tensor:
    """True at padded positions (funcodec/modules/nets_utils.py make_pad_mask)."""
    lengths = [int(v) for v in lengths]
    maxlen = max(lengths) if maxlen is None else maxlen
    ar = torch.arange(maxlen)[None, :]
    return ar >= torch.tensor(lengths)[:, None]


class Stack:
    """One rel-pos self-attention stack restated functionally: ConformerEncoder without CNN / macaron modules
    (conformer_encoder.py:537-616, EncoderLayer.forward :179-279) or TransformerEncoder_s0 (transformer_encoder.py:656-707,
    EncoderLayer.forward :92-163).  Both reduce to: embed; N x [x += MHA(LN(x)); x += FFN(LN(x))]; after_norm."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, spec):
        self.sd, self.p, self.s = sd, prefix, spec

    def w(self, name):
        return self.sd[f"{self.p}.{name}"]

    def embed(self, x: torch.Tensor):
        """Linear -> LayerNorm(eps 1e-5) -> [ReLU] -> RelPositionalEncoding (x * sqrt(d), pos_emb)
        (conformer_encoder.py:372-378; transformer_encoder.py:463-470)."""
        d = self.s.d_model
        x = F.linear(x, self.w("embed.0.weight"), self.w("embed.0.bias"))
        x = F.layer_norm(x, (d,), self.w("embed.1.weight"), self.w("embed.1.bias"), 1e-5)
        if self.s.embed_relu:
            x = torch.relu(x)
        x = x * math.sqrt(d)
        return x, rel_pos_emb(x.size(1), d)

    def attention(self, i: int, x: torch.Tensor, pos_emb: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """RelPositionMultiHeadedAttention.forward (attention.py:265-308) + forward_attention (:64-96); query = key = value."""
        h, d = self.s.heads, self.s.d_model
        dk = d // h
        pre = f"encoders.{i}.self_attn"
        nb = x.size(0)
        q = F.linear(x, self.w(f"{pre}.linear_q.weight"), self.w(f"{pre}.linear_q.bias")).view(nb, -1, h, dk)
        k = F.linear(x, self.w(f"{pre}.linear_k.weight"), self.w(f"{pre}.linear_k.bias")).view(nb, -1, h, dk).transpose(1, 2)
        v = F.linear(x, self.w(f"{pre}.linear_v.weight"), self.w(f"{pre}.linear_v.bias")).view(nb, -1, h, dk).transpose(1, 2)
        p = F.linear(pos_emb, self.w(f"{pre}.linear_pos.weight")).view(pos_emb.size(0), -1, h, dk).transpose(1, 2)
        q_u = (q + self.w(f"{pre}.pos_bias_u")).transpose(1, 2)
        q_v = (q + self.w(f"{pre}.pos_bias_v")).transpose(1, 2)
        ac = torch.matmul(q_u, k.transpose(-2, -1))
        bd = rel_shift(torch.matmul(q_v, p.transpose(-2, -1)))
        scores = (ac + bd) / math.sqrt(dk)
        m = mask.unsqueeze(1).eq(0)
        min_value = float(np.finfo(np.float32).min)
        scores = scores.masked_fill(m, min_value)
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(nb, -1, d)
        return F.linear(ctx, self.w(f"{pre}.linear_out.weight"), self.w(f"{pre}.linear_out.bias"))

    def ffn(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """PositionwiseFeedForward.forward (positionwise_feed_forward.py:33-35): w_2(act(w_1(x))); Swish = x * sigmoid(x)
        (nets_utils.py:568-574)."""
        pre = f"encoders.{i}.feed_forward"
        y = F.linear(x, self.w(f"{pre}.w_1.weight"), self.w(f"{pre}.w_1.bias"))
        y = torch.relu(y) if self.s.act == "relu" else y * torch.sigmoid(y)
        return F.linear(y, self.w(f"{pre}.w_2.weight"), self.w(f"{pre}.w_2.bias"))

    def trace(self, x: torch.Tensor, mask: torch.Tensor) -> dict:
        """Intermediates of forward() for the engine's debug probe (tools/laura_debug.py): keys (what, layer) like
        fc_laura_debug_probe -- 0 = stream after the input layer, 1 = attention-norm output, 2 = q/k/v stacked on the feature axis,
        3 = attention context before linear_out, 5 = residual stream at block entry (layer = n_blocks: before after_norm)."""
        h, d = self.s.heads, self.s.d_model
        n_att, n_ff = self.s.norm_names
        out = {}
        x, pos_emb = self.embed(x)
        out[(0, 0)] = x
        for i in range(self.s.layers):
            out[(5, i)] = x
            y = F.layer_norm(x, (d,), self.w(f"encoders.{i}.{n_att}.weight"), self.w(f"encoders.{i}.{n_att}.bias"), 1e-12)
            out[(1, i)] = y
            pre = f"encoders.{i}.self_attn"
            out[(2, i)] = torch.cat([F.linear(y, self.w(f"{pre}.linear_{k}.weight"), self.w(f"{pre}.linear_{k}.bias")) for k in "qkv"], -1)
            att = self.attention(i, y, pos_emb, mask)
            # context = linear_out^-1 is not available; recompute it the way attention() does
            out[(3, i)] = self._context(i, y, pos_emb, mask)
            x = x + att
            y = F.layer_norm(x, (d,), self.w(f"encoders.{i}.{n_ff}.weight"), self.w(f"encoders.{i}.{n_ff}.bias"), 1e-12)
            x = x + self.ffn(i, y)
        out[(5, self.s.layers)] = x
        return out

    def _context(self, i, x, pos_emb, mask):
        h, d = self.s.heads, self.s.d_model
        dk = d // h
        pre = f"encoders.{i}.self_attn"
        nb = x.size(0)
        q = F.linear(x, self.w(f"{pre}.linear_q.weight"), self.w(f"{pre}.linear_q.bias")).view(nb, -1, h, dk)
        k = F.linear(x, self.w(f"{pre}.linear_k.weight"), self.w(f"{pre}.linear_k.bias")).view(nb, -1, h, dk).transpose(1, 2)
        v = F.linear(x, self.w(f"{pre}.linear_v.weight"), self.w(f"{pre}.linear_v.bias")).view(nb, -1, h, dk).transpose(1, 2)
        p = F.linear(pos_emb, self.w(f"{pre}.linear_pos.weight")).view(pos_emb.size(0), -1, h, dk).transpose(1, 2)
        q_u = (q + self.w(f"{pre}.pos_bias_u")).transpose(1, 2)
        q_v = (q + self.w(f"{pre}.pos_bias_v")).transpose(1, 2)
        scores = (torch.matmul(q_u, k.transpose(-2, -1)) + rel_shift(torch.matmul(q_v, p.transpose(-2, -1)))) / math.sqrt(dk)
        m = mask.unsqueeze(1).eq(0)
        scores = scores.masked_fill(m, float(np.finfo(np.float32).min))
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
        return torch.matmul(attn, v).transpose(1, 2).contiguous().view(nb, -1, d)

    def forward(self, x: torch.Tensor, mask: torch.Tensor, return_layers: bool = False):
        """x [B, T, idim], mask [B, 1 or T, T] (True = attend).  LayerNorm eps 1e-12 (funcodec/modules/layer_norm.py:22)."""
        d = self.s.d_model
        n_att, n_ff = self.s.norm_names
        x, pos_emb = self.embed(x)
        layers = []
        for i in range(self.s.layers):
            r = x
            y = F.layer_norm(x, (d,), self.w(f"encoders.{i}.{n_att}.weight"), self.w(f"encoders.{i}.{n_att}.bias"), 1e-12)
            x = r + self.attention(i, y, pos_emb, mask)
            r = x
            y = F.layer_norm(x, (d,), self.w(f"encoders.{i}.{n_ff}.weight"), self.w(f"encoders.{i}.{n_ff}.bias"), 1e-12)
            x = r + self.ffn(i, y)
            if return_layers:
                layers.append(x)
        x = F.layer_norm(x, (d,), self.w("after_norm.weight"), self.w("after_norm.bias"), 1e-12)
        return (x, layers) if return_layers else x


class LauraOracle:
    def __init__(self, cfg: dict, state: Dict[str, np.ndarray]):
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from funcodec_amd.laura_config import laura_spec_from_config
        self.spec = s = laura_spec_from_config(cfg)
        self.sd = {k: _t(v).float() for k, v in state.items()}
        self.text_encoder = Stack(self.sd, "text_encoder", s.text_encoder)
        self.codec_lm = Stack(self.sd, "codec_lm.encoder", s.codec_lm)
        self.codec_encoder = Stack(self.sd, "codec_encoder", s.codec_encoder)
        self.K, self.D, self.nq = s.codebook_size, s.codebook_dim, s.predict_nq
        self.sos_eos, self.task_id = 0, 1

    # ---- text side --------------------------------------------------------------------------------------------------
    def token_embed(self, ids) -> torch.Tensor:
        """Text2Audio.tokenize_text (bin/text2audio_inference.py:99-113): model.token_embedding(token_idx)."""
        return F.embedding(_t(ids).long(), self.sd["token_embedding.weight"])

    def encode(self, text: torch.Tensor, text_lengths) -> torch.Tensor:
        """LauraGenModel.encode (laura_model.py:186-202): conformer text encoder + text_enc_out_layer.  text [B, L, idim]."""
        text = _t(text).float()
        masks = (~make_pad_mask(text_lengths, text.size(1)))[:, None, :]
        outs = self.text_encoder.forward(text, masks)
        return F.linear(outs, self.sd["text_enc_out_layer.weight"], self.sd["text_enc_out_layer.bias"])

    # ---- language model -----------------------------------------------------------------------------------------------
    def dense_vector(self, codec: torch.Tensor) -> torch.Tensor:
        """QuantizerCodebook.forward (laura_model.py:42-58) for full-length sequences: sum over the nq groups of
        embed[q][code]; codec [T, nq] -> [T, D]."""
        codec = _t(codec).long()
        nq = codec.shape[-1]
        shift = (1024 * torch.arange(32, dtype=torch.float32))[:nq].long()
        emb = self.sd["quantizer_codebook.embed"].reshape(-1, self.D)
        return F.embedding(codec + shift, emb).sum(dim=1)

    def llm_input(self, text_outs: torch.Tensor, codec: Optional[torch.Tensor]) -> torch.Tensor:
        """LauraGenModel.build_llm_io (laura_model.py:204-247), need_targets False, one utterance:
        [sos_eos, text, task_id, codec embeddings]; text_outs [L, D] (already cut to its length), codec [T, nq] or None."""
        lm_emb = self.sd["lm_embedding.weight"]
        parts = [lm_emb[self.sos_eos:self.sos_eos + 1], text_outs, lm_emb[self.task_id:self.task_id + 1]]
        if codec is not None and len(codec) > 0:
            parts.append(self.dense_vector(codec))
        return torch.cat(parts, dim=0)

    def lm_mask(self, T: int, input_length: int) -> torch.Tensor:
        """TransformerEmbedLM.score's mask (transformer_lm.py:283-288): causal, plus the block [0, input_length)^2 when
        bidirectional_inputs (input_length = 1 + text length: <sos> and the text, not the task id)."""
        m = torch.tril(torch.ones(T, T, dtype=torch.bool)).unsqueeze(0)
        if self.spec.bidirectional_inputs:
            m[:1, :input_length, :input_length] = True
        return m

    def lm_hidden(self, seq: torch.Tensor, input_length: int) -> torch.Tensor:
        return self.codec_lm.forward(seq.unsqueeze(0), self.lm_mask(seq.size(0), input_length))

    def lm_score(self, seq: torch.Tensor, input_length: int) -> torch.Tensor:
        """TransformerEmbedLM.score (transformer_lm.py:266-313), pe_type "uni", no cache: log-softmax of the decoder output at
        the LAST position; seq [T, D]."""
        h = self.lm_hidden(seq, input_length)
        y = F.linear(h[:, -1], self.sd["codec_lm.decoder.weight"], self.sd["codec_lm.decoder.bias"])[:, : self.spec.lm_vocab]
        return y.log_softmax(dim=-1).squeeze(0)

    def lm_score_all(self, seq: torch.Tensor, input_length: int) -> torch.Tensor:
        """log-softmax at EVERY position of one full-sequence pass ([T, V]).  Row t equals lm_score(seq[:t+1]) up to the
        rounding of differently blocked GEMMs (causal mask; the bidirectional block is inside every prefix that contains
        it), which is how the teacher-forced parity test of the engine reads all positions at once."""
        h = self.lm_hidden(seq, input_length)
        y = F.linear(h[0], self.sd["codec_lm.decoder.weight"], self.sd["codec_lm.decoder.bias"])[:, : self.spec.lm_vocab]
        return y.log_softmax(dim=-1)

    @staticmethod
    def sample_from(weighted_scores: torch.Tensor, sampling, u: float) -> int:
        """LauraGenModel.sampling_ids (laura_model.py:466-499) with the multinomial draw written as the inverse CDF of ONE
        uniform number u in [0, 1) (torch.multinomial consumes its own generator stream, which no second implementation can
        share; the distribution is the same).  sampling: False = greedy (topk(1)), True = full softmax, int = top-k,
        float = nucleus.  The candidate order is the reference's: index order / topk's descending order / stable descending sort."""
        if sampling is False:
            return int(weighted_scores.topk(1)[1][0])
        probs = weighted_scores.softmax(dim=0)
        if sampling is True:
            p, idx = probs, torch.arange(len(probs))
        elif isinstance(sampling, int):
            p, idx = probs.topk(sampling)
        else:
            sv, si = probs.sort(descending=True, stable=True)
            cum, n = 0.0, 0
            for i in range(len(si)):
                if cum < sampling:
                    cum += float(sv[i])
                    n += 1
                else:
                    break
            p, idx = sv[:n], si[:n]
        c = torch.cumsum(p.double(), 0)
        j = int(torch.searchsorted(c, torch.tensor(u * float(c[-1]), dtype=torch.float64), right=True))
        return int(idx[min(j, len(idx) - 1)])

    def decode_codec(self, text_outs: torch.Tensor, max_length: int, sampling=False, continual: Optional[List] = None,
                     uniforms: Optional[np.ndarray] = None, return_logp: bool = False):
        """LauraGenModel.decode_codec (laura_model.py:501-548) for one utterance: text_outs [L, D].  Returns tokens [T, nq]
        (and, optionally, the log-prob vector every step sampled from, [steps, V])."""
        out_tokens = [] if continual is None else [list(t) for t in continual]
        input_length = 1 + text_outs.size(0)
        eos = self.K + self.sos_eos
        logps = []
        for i in range(max_length):
            if len(out_tokens) > 0 and any(t == eos for t in out_tokens[-1]):
                break
            seq = self.llm_input(text_outs, torch.tensor(out_tokens, dtype=torch.int64) if out_tokens else None)
            pred = self.lm_score(seq, input_length)
            logps.append(pred)
            pred = pred.reshape(self.nq, -1)
            ids = []
            for k in range(self.nq):
                u = float(uniforms[i, k]) if uniforms is not None else 0.0
                ids.append(self.sample_from(pred[k], sampling, u))
            out_tokens.append(ids)
        if out_tokens and any(t == eos for t in out_tokens[-1]):
            out_tokens = out_tokens[:-1]
        toks = torch.tensor(out_tokens, dtype=torch.int64).reshape(-1, self.nq)
        return (toks, torch.stack(logps)) if return_logp else toks

    # ---- fine codec predictor -------------------------------------------------------------------------------------------
    def abs_pos(self, x: torch.Tensor) -> torch.Tensor:
        """PositionalEncoding.forward (embedding.py:79-91) of width codebook_dim: x * sqrt(D) + pe[:len]; x [T, D]."""
        return x * math.sqrt(self.D) + sinusoid_table(x.size(0), self.D)

    def cal_codec_emb(self, text_outs: List[torch.Tensor], codecs: List[torch.Tensor]) -> List[torch.Tensor]:
        """LauraGenModel.cal_codec_emb (laura_model.py:296-333) fed with the one-hot probabilities of syn_audio (:558-563):
        the matmul with a one-hot row is that codebook row, summed over the predict_nq groups in group order.  Lists over
        the batch: text_outs[i] [L_i, D], codecs[i] [T_i, >= nq].  Returns [T_i, D] per utterance."""
        ins, lens = [], []
        for t, c in zip(text_outs, codecs):
            c = torch.clamp(_t(c).long()[:, : self.nq], 0, self.K - 1)
            first = None
            for k in range(self.nq):
                one = F.embedding(c[:, k], self.sd["quantizer_codebook.embed"][k])
                first = one if first is None else first + one
            if self.spec.pos_emb_type == "split":
                one_in = [self.abs_pos(t), self.abs_pos(first)]
            else:
                one_in = [t, first]
            ins.append(torch.cat(one_in, dim=0))
            lens.append(ins[-1].size(0))
        T = max(lens)
        x = torch.stack([F.pad(v, (0, 0, 0, T - v.size(0))) for v in ins])
        masks = (~make_pad_mask(lens, T))[:, None, :]
        outs = self.codec_encoder.forward(x, masks)
        outs = F.linear(outs, self.sd["codec_encoder_out_layer.weight"], self.sd["codec_encoder_out_layer.bias"])
        return [outs[i, t.size(0): t.size(0) + c.size(0)] for i, (t, c) in enumerate(zip(text_outs, codecs))]
