"""MED (BERT with cross-attention) on the HIP kernels.

Mirror of the parts of the reference's models/med.py that the hot path executes:
``BertModel`` (ITM text encoder, models/blip_itm.py:31) and ``BertLMHeadModel``
(caption decoder, models/blip.py:98).  Parameter names are the reference's, so
BLIP checkpoints load unchanged (``bert.encoder.layer.N.crossattention.self.key.weight`` …).

Schedule differences from the reference (results are the same function):
  * cross-attention K/V are projected ONCE per image per layer
    (``project_cross_kv``) instead of on every call for every beam
    (models/med.py:160-163);
  * the decoder KV cache is a preallocated [rows, H, max_len, 64] / V^T buffer the
    GEMM epilogue appends into, instead of ``torch.cat`` per step (:164-168);
  * q/k/v of self-attention are one fused GEMM.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from . import kernels as K
from .packing import FP8, PackedCache, fold_layernorm, parity_attention_arith, parity_attention_f32, parity_attention_kind, require_cuda, v32, w3, w8, w16

LN_EPS_DEFAULT = 1e-12


class BertConfig:
    """The fields of configs/med_config.json the hot path reads."""

    def __init__(self, **kw):
        self.hidden_size = 768
        self.num_hidden_layers = 12
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.hidden_act = "gelu"
        self.layer_norm_eps = LN_EPS_DEFAULT
        self.max_position_embeddings = 512
        self.vocab_size = 30524
        self.pad_token_id = 0
        self.type_vocab_size = 2
        self.initializer_range = 0.02
        self.encoder_width = 768
        self.add_cross_attention = True
        self.hidden_dropout_prob = 0.1
        self.attention_probs_dropout_prob = 0.1
        for k, v in kw.items():
            setattr(self, k, v)
        if self.hidden_act != "gelu":
            raise ValueError("only hidden_act='gelu' (erf) is implemented")
        if self.hidden_size // self.num_attention_heads != 64:
            raise ValueError("vidil_amd MED kernels are built for head_dim 64")

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


# ---- parameter holders (names = reference state_dict keys) -------------------------
class _SelfAttn(nn.Module):
    def __init__(self, cfg, cross):
        super().__init__()
        kv_in = cfg.encoder_width if cross else cfg.hidden_size
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.key = nn.Linear(kv_in, cfg.hidden_size)
        self.value = nn.Linear(kv_in, cfg.hidden_size)


class _DenseLN(nn.Module):
    def __init__(self, d_in, d_out, eps):
        super().__init__()
        self.dense = nn.Linear(d_in, d_out)
        self.LayerNorm = nn.LayerNorm(d_out, eps=eps)


class _Attention(nn.Module):
    def __init__(self, cfg, cross=False):
        super().__init__()
        self.self = _SelfAttn(cfg, cross)
        self.output = _DenseLN(cfg.hidden_size, cfg.hidden_size, cfg.layer_norm_eps)


class _Intermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)


class BertLayer(nn.Module):
    def __init__(self, cfg, layer_num):
        super().__init__()
        self.attention = _Attention(cfg)
        if cfg.add_cross_attention:
            self.crossattention = _Attention(cfg, cross=True)
        self.intermediate = _Intermediate(cfg)
        self.output = _DenseLN(cfg.intermediate_size, cfg.hidden_size, cfg.layer_norm_eps)


class BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(cfg, i) for i in range(cfg.num_hidden_layers)])


class BertEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(cfg.max_position_embeddings).expand((1, -1)))


def _init_bert(module, std):
    """models/med.py:558-568."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=std)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


class CrossKV:
    """Per-image cross-attention keys / values^T of every layer: K [L][B,H,Te,64], VT [L][B,H,64,NP]."""

    def __init__(self, k, vt, B, Te, NP, last_vt=None, last_NP=0, tiled=False, Tk_cap=None, f32=False):
        self.k, self.vt, self.B, self.Te, self.NP = k, vt, B, Te, NP
        # f32 (parity precision mode with vidil_attention_f32): k = f32 [L][B, Te, 2C] — the K | V projection GEMM's row-major
        # output, keys in columns 0..C-1 and values in C..2C-1 — and vt a placeholder [L][B, 1] (same batch axis for adopt())
        self.f32 = f32
        # tiled: K and V of every layer in 32-key fragment tiles [L][B,H,Tk_cap/32,2048] (vidil_attention kv_tiled)
        self.tiled, self.Tk_cap = tiled, (Te if Tk_cap is None else Tk_cap)
        # optional second copy of the LAST layer's values in V^T layout (see project_cross_kv(last_layer_vt=True))
        self.last_vt, self.last_NP = last_vt, last_NP


class BeamArena:
    """Append-only self-attention K/V of a beam search (all layers): k, v f16 [L][Tcap][rows][H*64], written at
    [position][slot = producing beam row]; ``anc`` i32 [rows][Tcap] maps (beam row, position) -> slot and is the
    only thing reordered per step (vidil_beam_ancestry) — the reference's _reorder_cache without moving the cache."""

    def __init__(self, L, Tcap, rows, C, device, dtype=torch.float16):
        self.L, self.Tcap, self.rows = L, Tcap, rows
        self.k = torch.empty((L, Tcap, rows, C), dtype=dtype, device=device)
        self.v = torch.empty((L, Tcap, rows, C), dtype=dtype, device=device)
        self._anc = [torch.zeros((rows, Tcap), dtype=torch.int32, device=device) for _ in range(2)]
        self._cur = 0

    @property
    def anc(self):
        return self._anc[self._cur]

    def init_prompt(self, P, group):
        """Positions 0..P-1 of beam row r live in slot (r // group) * group (group = nb after a shared prompt
        pass over one row per image, 1 when every row ran its own prompt pass)."""
        r = torch.arange(self.rows, dtype=torch.int32, device=self.k.device)
        self.anc[:, :P] = (torch.div(r, group, rounding_mode="floor") * group).to(torch.int32)[:, None]

    def reorder(self, beam_idx_i32, cur_pos):
        """Beam row r now continues old row beam_idx[r]; its next K/V (position cur_pos) go to slot r."""
        K.beam_ancestry(self._anc[self._cur], self._anc[self._cur ^ 1], beam_idx_i32, cur_pos)
        self._cur ^= 1


class BertModel(PackedCache, nn.Module):
    """ITM text encoder / decoder trunk.  ``forward`` is not the generic HF signature: the hot
    path calls ``encode`` (ITM) or is driven by ``BertLMHeadModel``."""

    #: images per cross-attention / cross K|V projection launch of a decode batch (a search over more images — the images of
    #: several tower chunks — issues these launches per block of this many images: byte-floor kernels whose launch already fills
    #: the chip, kept at the sizes every per-unit offset of theirs has been exercised at)
    MAX_IMAGES_PER_LAUNCH = 4096

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        if add_pooling_layer:
            raise ValueError("the pooler is not on the hot path (reference builds add_pooling_layer=False)")
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.apply(lambda m: _init_bert(m, config.initializer_range))

    # ------------------------------------------------------------------ packing
    def _pack(self):
        e = self.embeddings
        c = self.cdt
        p = dict(word=v32(e.word_embeddings.weight).view(self.config.vocab_size, -1),
                 pos=v32(e.position_embeddings.weight).view(self.config.max_position_embeddings, -1),
                 emb_g=v32(e.LayerNorm.weight), emb_b=v32(e.LayerNorm.bias), layers=[])
        for l in self.encoder.layer:
            a, o = l.attention.self, l.attention.output
            d = dict(qkv_w=w16(a.query.weight, a.key.weight, a.value.weight, dtype=c),
                     qkv_b=v32(a.query.bias, a.key.bias, a.value.bias),
                     ao_w=w16(o.dense.weight, dtype=c), ao_b=v32(o.dense.bias),
                     ao_g=v32(o.LayerNorm.weight), ao_bt=v32(o.LayerNorm.bias),
                     i_w=w16(l.intermediate.dense.weight, dtype=c), i_b=v32(l.intermediate.dense.bias),
                     o_w=w16(l.output.dense.weight, dtype=c), o_b=v32(l.output.dense.bias),
                     o_g=v32(l.output.LayerNorm.weight), o_bt=v32(l.output.LayerNorm.bias))
            if hasattr(l, "crossattention"):
                ca, co = l.crossattention.self, l.crossattention.output
                d.update(cq_w=w16(ca.query.weight, dtype=c), cq_b=v32(ca.query.bias),
                         ckv_w=w16(ca.key.weight, ca.value.weight, dtype=c), ckv_b=v32(ca.key.bias, ca.value.bias),
                         co_w=w16(co.dense.weight, dtype=c), co_b=v32(co.dense.bias),
                         co_g=v32(co.LayerNorm.weight), co_bt=v32(co.LayerNorm.bias))
                if self.fp8:     # fp8 tower mode: the image-side K|V projection (its A rows are a tower's output) on e4m3 too
                    d["ckv_w8"], d["ckv_s"] = w8(ca.key.weight, ca.value.weight)
            if self.parity:      # parity precision mode: [W_hi | W_hi | W_lo] for every GEMM of the stack
                d.update(qkv_w3=w3(a.query.weight, a.key.weight, a.value.weight, dtype=c), ao_w3=w3(o.dense.weight, dtype=c),
                         i_w3=w3(l.intermediate.dense.weight, dtype=c), o_w3=w3(l.output.dense.weight, dtype=c))
                if hasattr(l, "crossattention"):
                    d.update(cq_w3=w3(ca.query.weight, dtype=c), ckv_w3=w3(ca.key.weight, ca.value.weight, dtype=c),
                             co_w3=w3(co.dense.weight, dtype=c))
            p["layers"].append(d)
        p["parity"] = self.parity
        return p

    def pack_flags(self):
        # (the attention kind of the parity mode decides the dtype of a session's KV arena and cross K|V: a change re-packs, and
        #  with the new pack object every DecoderSession / captured step graph keyed on it is rebuilt — ADVICE r4)
        return (self.parity, parity_attention_kind(self) if self.parity else None)

    def _folded(self, p):
        """Weights of the LN-folded text stack (built on first use, kept with the pack): per layer the cross query and
        fc1 against the LayerNorms of the SAME layer's self / cross blocks, and (from layer 1 on) Q|K|V against the
        previous layer's output LayerNorm — W' = T16(gamma (.) W), b' = b + W·beta, colsum (packing.fold_layernorm)."""
        if "folded" in p:
            return p["folded"]
        c = self.cdt
        out = []
        layers = list(self.encoder.layer)
        for i, l in enumerate(layers):
            d = {}
            if hasattr(l, "crossattention"):
                ln1, ln2 = l.attention.output.LayerNorm, l.crossattention.output.LayerNorm
                ca = l.crossattention.self
                d["cq"] = fold_layernorm(ca.query.weight, ca.query.bias, ln1.weight, ln1.bias, c)
                d["fc1"] = fold_layernorm(l.intermediate.dense.weight, l.intermediate.dense.bias, ln2.weight, ln2.bias, c)
            if i > 0:
                a, ln3 = l.attention.self, layers[i - 1].output.LayerNorm
                d["qkv"] = fold_layernorm(torch.cat([a.query.weight, a.key.weight, a.value.weight]),
                                          torch.cat([a.query.bias, a.key.bias, a.value.bias]), ln3.weight, ln3.bias, c)
            out.append(d)
        p["folded"] = out
        return out

    # --------------------------------------------------------- cross K/V (once per image)
    def project_cross_kv(self, enc16, B, Te, out: "CrossKV" = None, v_rowmajor=False, last_layer_vt=False, tiled=False):
        """enc16: f16 [B*Te, encoder_width] image tokens.  One fused K|V GEMM per layer.  ``out``: buffers of a
        previous call with the same (B, Te) to overwrite (keeps device addresses stable for captured graphs).
        ``v_rowmajor``: keep V as [L][B,H,Te,64] (NP = 0) — for consumers whose every cross-attention launch has more
        than 32 query rows per image (the staged kernel transposes in LDS); the decode steps need V^T.
        ``last_layer_vt`` (with v_rowmajor): the last layer's V goes to a V^T buffer instead (``last_vt``) — its
        consumer is encode_cls, whose last layer has one query row per pair.
        ``tiled``: K and V in 32-key fragment tiles — for consumers whose every launch has at most 32 query rows per
        image and which re-read the K/V from HBM many times (the decode steps of the captioner)."""
        p = self.packed()
        H = self.config.num_attention_heads
        cdt = enc16.dtype
        if p["parity"]:
            # parity precision mode: enc16 holds [hi | lo | hi] rows of the image tokens (the ViT's parity output)
            if enc16.shape[1] != 3 * self.config.encoder_width:
                raise K.VidilHipError(f"project_cross_kv (parity mode): image tokens must be [hi | lo | hi] rows of width "
                                      f"{3 * self.config.encoder_width}, got {tuple(enc16.shape)}")

            # (split-operand attention: consumers with at most 32 query rows per image — the caption decoder — keep the 16-bit
            #  fragment tiles of the plain path, written by the K-tripled GEMM's heads epilogue: vidil_attention_f32 kv16)
            if parity_attention_f32(self) and not (tiled and parity_attention_arith(self) == 1):
                L, C = len(p["layers"]), self.config.hidden_size
                dev = enc16.device
                if out is not None and getattr(out, "f32", False) and (out.B, out.Te) == (B, Te) and out.k.device == dev:
                    kv32, ph = out.k, out.vt
                else:
                    kv32 = torch.empty((L, B, Te, 2 * C), dtype=torch.float32, device=dev)
                    ph = torch.zeros((L, B, 1), dtype=torch.float32, device=dev)
                for i, d in enumerate(p["layers"]):
                    K.gemm(enc16, d["ckv_w3"], d["ckv_b"], out=kv32[i].view(B * Te, 2 * C), split_k=True)
                return CrossKV(kv32, ph, B, Te, 0, f32=True)

            def kv_gemm(d, r0=0, r1=None, **heads):
                K.gemm(enc16[r0:r1], d["ckv_w3"], d["ckv_b"], heads=heads, split_k=True)
        elif self.fp8 and "ckv_w8" in p["layers"][0] and enc16.shape[1] % 128 == 0:
            # fp8 tower mode (BASELINE config 5): e4m3 image tokens x e4m3 K|V weights, K / V written in the 16-bit type
            # (clamped first: torch's e4m3 cast does not saturate — 470 becomes NaN — while every device-side conversion does)
            enc8 = enc16.clamp(-448.0, 448.0).to(FP8)

            def kv_gemm(d, r0=0, r1=None, **heads):
                K.gemm(enc8[r0:r1], d["ckv_w8"], d["ckv_b"], w_scale=d["ckv_s"], dtype16=cdt, heads=heads)
        else:
            def kv_gemm(d, r0=0, r1=None, **heads):
                K.gemm(enc16[r0:r1], d["ckv_w"], d["ckv_b"], heads=heads)
        if tiled:
            Tc = (Te + 31) // 32 * 32
            L = len(p["layers"])
            dev = enc16.device
            if out is not None and out.tiled and (out.B, out.Te) == (B, Te) and out.k.device == dev:
                k, v = out.k, out.vt
            else:
                k = torch.empty((L, B, H, Tc, 64), dtype=cdt, device=dev)
                v = torch.empty((L, B, H, Tc, 64), dtype=cdt, device=dev)
            # (the images of several tower chunks — a decode batch, round 6 — are projected MAX_IMAGES_PER_LAUNCH at a time: the
            #  launch sizes the tower-sized path has always run, whatever the search holds)
            for i, d in enumerate(p["layers"]):
                for b0 in range(0, B, self.MAX_IMAGES_PER_LAUNCH):
                    b1 = min(B, b0 + self.MAX_IMAGES_PER_LAUNCH)
                    kv_gemm(d, b0 * Te, b1 * Te, k=k[i][b0:b1], vt=v[i][b0:b1], T=Te, H=H, part0=1, t_off=0, Tk_cap=Tc, tiled=True)
            return CrossKV(k, v, B, Te, Tc, tiled=True, Tk_cap=Tc)
        NPt = (Te + 15) // 16 * 16
        NP = 0 if v_rowmajor else NPt
        L = len(p["layers"])
        dev = enc16.device
        if out is not None and not out.tiled and (out.B, out.Te, out.NP) == (B, Te, NP) and out.k.device == dev:
            k, vt = out.k, out.vt
        else:
            k = torch.empty((L, B, H, Te, 64), dtype=cdt, device=dev)
            vt = torch.empty((L, B, H, Te, 64) if v_rowmajor else (L, B, H, 64, NP), dtype=cdt, device=dev)
        last_vt = None
        if v_rowmajor and last_layer_vt:
            last_vt = torch.empty((B, H, 64, NPt), dtype=cdt, device=dev)
        for i, d in enumerate(p["layers"]):
            if last_vt is not None and i == L - 1:
                kv_gemm(d, k=k[i], vt=last_vt, T=Te, H=H, part0=1, t_off=0, Tk_cap=Te, NP=NPt)
            else:
                kv_gemm(d, k=k[i], vt=vt[i], T=Te, H=H, part0=1, t_off=0, Tk_cap=Te, NP=NP)
        return CrossKV(k, vt, B, Te, NP, last_vt, NPt if last_vt is not None else 0)

    # ------------------------------------------------------------------ layers
    def run_layers(self, h32, h16, *, rows, T, self_k, self_vt, t_off, Tk_cap, NPs, causal, kv_len,
                   cross: CrossKV, cross_index=None, cross_group=1, cross_groups=None, cross_max_group=0, ws=None,
                   arena: "BeamArena" = None, arena_slot_stride=1, n_layers=None, self_done_first=False,
                   stop_after_self=False, fused=None):
        """Run every layer on the f32/f16 hidden pair (both [rows*T, C], updated in place).

        self_k / self_vt: [L][rows,H,Tk_cap,64] / [L][rows,H,64,NPs] — this call's keys are appended at
        ``t_off`` and attention runs over t_off+T keys (causal inside the new block when ``causal``).

        arena (beam-search decoding): with T == 1 the new key/value of row r is appended at
        arena[position t_off][slot r] and attention follows the row's ancestry (self_k / self_vt unused);
        with T > 1 (prompt pass, t_off == 0) attention runs over the block as above and the block's K/V are
        ALSO written to the arena at slots r*arena_slot_stride.
        """
        p = self.packed()
        cfg = self.config
        H, C = cfg.num_attention_heads, cfg.hidden_size
        eps = cfg.layer_norm_eps
        M = rows * T
        dev = h32.device
        cdt = h16.dtype
        if p["parity"]:
            if self_done_first or stop_after_self or n_layers is not None:
                raise K.VidilHipError("run_layers: the parity precision mode runs whole stacks (the caption decoder, "
                                      "BertModel.encode); encode_cls' split schedules are not built for it — BLIP_ITM.itm_pairs "
                                      "takes the encode() route in that mode")
            return self._run_layers_parity(p, h32, h16, rows=rows, T=T, self_k=self_k, self_vt=self_vt, t_off=t_off,
                                           Tk_cap=Tk_cap, NPs=NPs, causal=causal, kv_len=kv_len, cross=cross,
                                           cross_index=cross_index, cross_group=cross_group, cross_groups=cross_groups,
                                           cross_max_group=cross_max_group, ws=ws, arena=arena, arena_slot_stride=arena_slot_stride)
        if fused is None:
            # encoder batches with cross-attention (the ITM pairs) run without LayerNorm launches, WHATEVER their size: a
            # pair's logits must not depend on how many other pairs share its batch (ranks / tail batches of different
            # sizes write the same JSON), so the choice cannot depend on M ($VIDIL_FUSE_LN_MIN_ROWS is for experiments).
            # Decoder sessions (arena) state their choice themselves (DecoderSession.fused_ln), for the same reason.
            fused = (self._text_fold_ok(cdt) and cross is not None and arena is None and not stop_after_self
                     and M >= int(os.environ.get("VIDIL_FUSE_LN_MIN_ROWS", 0)))
        elif fused and not (self._text_fold_ok(cdt) and cross is not None and not stop_after_self):
            raise K.VidilHipError("run_layers(fused=True): the LN-folded text stack needs cross-attention layers, 16-bit operands and "
                                  "hidden_size % 64 == 0, <= 1024")
        if fused:
            return self._run_layers_fused(h32, h16, rows=rows, T=T, self_k=self_k, self_vt=self_vt, t_off=t_off, Tk_cap=Tk_cap,
                                          NPs=NPs, causal=causal, kv_len=kv_len, cross=cross, cross_index=cross_index,
                                          cross_group=cross_group, cross_groups=cross_groups, cross_max_group=cross_max_group,
                                          n_layers=n_layers, self_done_first=self_done_first, arena=arena,
                                          arena_slot_stride=arena_slot_stride)
        if ws is None:
            ws = {}
        q = ws.get("q")
        if q is None or q.shape[0] < rows or q.shape[2] != T:
            q = torch.empty((rows, H, T, 64), dtype=cdt, device=dev)
            ws["q"] = q
        o = torch.empty((M, C), dtype=cdt, device=dev)
        tmp = torch.empty((M, C), dtype=torch.float32, device=dev)
        inter = torch.empty((M, cfg.intermediate_size), dtype=cdt, device=dev)
        Nk = t_off + T
        if arena is not None and T > 1 and t_off != 0:
            raise K.VidilHipError("run_layers: a multi-token block can only be appended to a beam arena at position 0")
        # n_layers: only the first n (encode_cls runs the last itself).  self_done_first / stop_after_self split layer
        # 0 after its self-attention block (dense + residual + LayerNorm included): that block does not see the image,
        # so encode_cls runs it once per TEXT and the rest of the stack once per (image, text) pair.
        for i, d in enumerate(p["layers"][:n_layers]):
            if self_done_first and i == 0:
                pass
            elif arena is not None and T == 1:
                K.gemm(h16, d["qkv_w"], d["qkv_b"],
                       arena=dict(q=q, k=arena.k[i], v=arena.v[i], T=1, H=H, part0=0, t_off=t_off, Tcap=arena.Tcap,
                                  arena_rows=arena.rows, slot_stride=1, q_scale=0.125))
                K.beam_attention(q, arena.k[i], arena.v[i], arena.anc, o, rows=rows, H=H, n_keys=Nk)
            else:
                K.gemm(h16, d["qkv_w"], d["qkv_b"],
                       heads=dict(q=q, k=self_k[i], vt=self_vt[i], T=T, H=H, part0=0, t_off=t_off, Tq_cap=T,
                                  Tk_cap=Tk_cap, NP=NPs, q_scale=0.125))
                K.attention(q, self_k[i], self_vt[i], o, Bq=rows, H=H, Nq=T, Nk=Nk, Tq_cap=T, Tk_cap=Tk_cap, NP=NPs,
                            causal=causal, causal_off=t_off, kv_len=kv_len)
                if arena is not None:   # the prompt's K/V, once more, in the arena layout (a prompt is a few tokens)
                    K.gemm(h16, d["qkv_w"][C:], d["qkv_b"][C:],
                           arena=dict(k=arena.k[i], v=arena.v[i], T=T, H=H, part0=1, t_off=0, Tcap=arena.Tcap,
                                      arena_rows=arena.rows, slot_stride=arena_slot_stride))
            if not (self_done_first and i == 0):
                K.gemm(o, d["ao_w"], d["ao_b"], out=tmp, resid=h32)
                K.layernorm(tmp, d["ao_g"], d["ao_bt"], eps, out16=h16, out32=h32)
            if stop_after_self:
                break
            if cross is not None:
                # every query batch that shares an image (the beams of a caption search, the captions of a
                # frame) is served by one fetch of that image's K/V: see vidil_attention's grouping forms
                K.gemm(h16, d["cq_w"], d["cq_b"], heads=dict(q=q, T=T, H=H, part0=0, Tq_cap=T, q_scale=0.125))
                for b0, b1 in self._cross_blocks(cross, cross_index, cross_groups):
                    r0, r1 = (0, rows) if (b0, b1) == (0, cross.B) else (b0 * cross_group, b1 * cross_group)
                    K.attention(q[r0:r1], cross.k[i][b0:b1], cross.vt[i][b0:b1], o[r0 * T:r1 * T], Bq=r1 - r0, H=H, Nq=T, Nk=cross.Te,
                                Tq_cap=T, Tk_cap=cross.Tk_cap, NP=cross.NP, kv_group=cross_group, kv_index=cross_index,
                                group_start=cross_groups, max_group=cross_max_group, kv_tiled=cross.tiled)
                K.gemm(o, d["co_w"], d["co_b"], out=tmp, resid=h32)
                K.layernorm(tmp, d["co_g"], d["co_bt"], eps, out16=h16, out32=h32)
            K.gemm(h16, d["i_w"], d["i_b"], out=inter, act=K.ACT_GELU_ERF)
            K.gemm(inter, d["o_w"], d["o_b"], out=tmp, resid=h32)
            K.layernorm(tmp, d["o_g"], d["o_bt"], eps, out16=h16, out32=h32)
        return h32, h16

    def _cross_blocks(self, cross, cross_index, cross_groups):
        """Image ranges [b0, b1) of the cross-attention launches of one layer: the whole batch, or — uniform grouping (a beam
        search: image b serves query batches b * group .. b * group + group - 1) over more than MAX_IMAGES_PER_LAUNCH images —
        blocks of that many images, each a launch over contiguous slices of Q, K, V and the output rows."""
        B, n = cross.B, self.MAX_IMAGES_PER_LAUNCH
        if cross_index is not None or cross_groups is not None or B <= n:
            return [(0, B)]                           # (one launch over every query row, whatever maps rows to images)
        return [(b0, min(B, b0 + n)) for b0 in range(0, B, n)]

    def _run_layers_parity(self, p, h32, h3, *, rows, T, self_k, self_vt, t_off, Tk_cap, NPs, causal, kv_len, cross, cross_index,
                           cross_group, cross_groups, cross_max_group, ws, arena, arena_slot_stride):
        """run_layers in the parity precision mode: ``h3`` [rows*T, 3C] carries the hidden states as [hi | lo | hi]
        operand rows, every GEMM runs against [W_hi | W_hi | W_lo] with K tripled, LayerNorm / attention write split rows
        directly and the GELU output goes through f32 + vidil_split3_f32.  Same launch sequence otherwise."""
        cfg = self.config
        H, C = cfg.num_attention_heads, cfg.hidden_size
        eps = cfg.layer_norm_eps
        M = rows * T
        dev, cdt = h32.device, h3.dtype
        if ws is None:
            ws = {}
        q = ws.get("q")
        if q is None or q.shape[0] < rows or q.shape[2] != T:
            q = torch.empty((rows, H, T, 64), dtype=cdt, device=dev)
            ws["q"] = q
        o3 = torch.empty((M, 3 * C), dtype=cdt, device=dev)
        tmp = torch.empty((M, C), dtype=torch.float32, device=dev)
        inter3 = torch.empty((M, 3 * cfg.intermediate_size), dtype=cdt, device=dev)
        # Which producers may leave the third plane of their [hi | lo | hi] rows unwritten: only those whose EVERY
        # consumer takes the K-loop form of the compensated product, asked per CALL (K.split_k_serves: the process-wide switch
        # is necessary, not sufficient).  inter3 feeds fc2 (f32 epilogue); with the f32-row attention kinds h3 / o3 feed Q|K|V,
        # the cross query, both output projections and fc1, all with the f32 epilogue; kind "16" feeds h3 to arena / short-sequence
        # scatter epilogues, which run the plain K = 3 Kl product over all three planes.  The consumers state a_planes, so a
        # launch that would read an unwritten plane fails (EINVAL) instead of computing on it.
        f32_attn, arith = parity_attention_f32(self), parity_attention_arith(self)
        qkv32 = torch.empty((M, 3 * C), dtype=torch.float32, device=dev) if f32_attn else None
        q32 = torch.empty((M, C), dtype=torch.float32, device=dev) if f32_attn and cross is not None else None
        d0 = p["layers"][0]
        planes = planes_h = 3
        if K.split_k_in_loop():
            if K.split_k_serves(inter3, d0["o_w3"], d0["o_b"], out=tmp, resid=h32):
                planes = 2
            if f32_attn and planes == 2 and all((
                    K.split_k_serves(h3, d0["qkv_w3"], d0["qkv_b"], out=qkv32),
                    K.split_k_serves(o3, d0["ao_w3"], d0["ao_b"], out=tmp, resid=h32),
                    K.split_k_serves(h3, d0["i_w3"], d0["i_b"], split3_out=inter3, act=K.ACT_GELU_ERF),
                    cross is None or K.split_k_serves(h3, d0["cq_w3"], d0["cq_b"], out=q32))):
                planes_h = 2
        if planes == 2 and os.environ.get("VIDIL_POISON_SPLIT3") == "1":       # (developer: NaNs in the unwritten third planes —
            for _b in (inter3,):                               #  any consumer that reads one shows up at once)
                _b[:, 2 * (_b.shape[1] // 3):] = float("nan")
        if planes_h == 2 and os.environ.get("VIDIL_POISON_SPLIT3") == "1":       # (developer: NaNs in the unwritten third planes —
            for _b in (o3, h3,):                               #  any consumer that reads one shows up at once)
                _b[:, 2 * (_b.shape[1] // 3):] = float("nan")
        Nk = t_off + T
        if arena is not None and T > 1 and t_off != 0:
            raise K.VidilHipError("run_layers: a multi-token block can only be appended to a beam arena at position 0")
        if f32_attn:
            # Q | K | V (and the cross query) stay f32 and row-major, the KV arena and the cross K | V are f32:
            # vidil_attention_f32 reads all of them in place.  (t_off > 0 happens in the arena form only.)
            # (split-operand kind: the cross K | V of a decoder session are 16-bit fragment tiles instead)
            kv16 = cross is not None and arith == 1 and cross.tiled and not getattr(cross, "f32", False)
            if (arena is not None and arena.k.dtype != torch.float32) or (cross is not None and not getattr(cross, "f32", False) and not kv16):
                raise K.VidilHipError("run_layers (parity mode, f32 attention): the KV arena / cross K|V of this session were built for "
                                      "the 16-bit attention kernels — build the DecoderSession / CrossKV under the same $VIDIL_PARITY_ATTN")
            if t_off != 0 and not (arena is not None and T == 1):
                raise K.VidilHipError("run_layers (parity mode, f32 attention): cached self-attention keys are served by the arena form only")
            if kv16 and T * max(1, cross_group) > 32 and cross_groups is None and cross_index is None:
                # (the kv16 form serves at most 32 query rows per image — a DecoderSession's decode steps and shared
                #  prompt pass; a longer prompt x group must be built with the f32-row cross K|V instead)
                raise K.VidilHipError(f"run_layers (parity mode, split attention): {T} tokens x {cross_group} sequences per image exceed the 32 "
                                      "query rows per image the 16-bit cross K/V tiles serve — build the session with tiled_cross=False")
        for i, d in enumerate(p["layers"]):
            if f32_attn:
                K.gemm(h3, d["qkv_w3"], d["qkv_b"], out=qkv32, split_k=True, a_planes=planes_h)
                if arena is not None and T == 1:
                    arena.k[i][t_off].copy_(qkv32[:, C:2 * C])          # position t_off, slot = producing beam row
                    arena.v[i][t_off].copy_(qkv32[:, 2 * C:])
                    K.attention_f32(qkv32[:, :C], arena.k[i], arena.v[i], o3, Bq=rows, H=H, Nq=1, Nk=Nk, anc=arena.anc,
                                    arena_rows=arena.rows, planes=planes_h)
                else:
                    K.attention_f32(qkv32[:, :C], qkv32[:, C:2 * C], qkv32[:, 2 * C:], o3, Bq=rows, H=H, Nq=T, Nk=T, causal=causal,
                                    kv_len=kv_len, arith=arith, planes=planes_h)
                    if arena is not None:   # the prompt's K / V in the arena: position t, slot r * arena_slot_stride
                        blk = qkv32.view(rows, T, 3 * C)
                        arena.k[i][:T, 0:rows * arena_slot_stride:arena_slot_stride] = blk[:, :, C:2 * C].permute(1, 0, 2)
                        arena.v[i][:T, 0:rows * arena_slot_stride:arena_slot_stride] = blk[:, :, 2 * C:].permute(1, 0, 2)
            elif arena is not None and T == 1:
                K.gemm(h3, d["qkv_w3"], d["qkv_b"], split_k=True,
                       arena=dict(q=q, k=arena.k[i], v=arena.v[i], T=1, H=H, part0=0, t_off=t_off, Tcap=arena.Tcap,
                                  arena_rows=arena.rows, slot_stride=1, q_scale=0.125))
                K.beam_attention(q, arena.k[i], arena.v[i], arena.anc, o3, rows=rows, H=H, n_keys=Nk, split3=True)
            else:
                K.gemm(h3, d["qkv_w3"], d["qkv_b"], split_k=True,
                       heads=dict(q=q, k=self_k[i], vt=self_vt[i], T=T, H=H, part0=0, t_off=t_off, Tq_cap=T,
                                  Tk_cap=Tk_cap, NP=NPs, q_scale=0.125))
                K.attention(q, self_k[i], self_vt[i], o3, Bq=rows, H=H, Nq=T, Nk=Nk, Tq_cap=T, Tk_cap=Tk_cap, NP=NPs,
                            causal=causal, causal_off=t_off, kv_len=kv_len, split3=True)
                if arena is not None:
                    K.gemm(h3, d["qkv_w3"][C:], d["qkv_b"][C:], split_k=True,
                           arena=dict(k=arena.k[i], v=arena.v[i], T=T, H=H, part0=1, t_off=0, Tcap=arena.Tcap,
                                      arena_rows=arena.rows, slot_stride=arena_slot_stride))
            K.gemm(o3, d["ao_w3"], d["ao_b"], out=tmp, resid=h32, split_k=True, a_planes=planes_h)
            K.layernorm(tmp, d["ao_g"], d["ao_bt"], eps, out16=h3, out32=h32, split3=True, planes=planes_h)
            if cross is not None and f32_attn:
                K.gemm(h3, d["cq_w3"], d["cq_b"], out=q32, split_k=True, a_planes=planes_h)
                if kv16:
                    for b0, b1 in self._cross_blocks(cross, cross_index, cross_groups):
                        r0, r1 = (0, M) if (b0, b1) == (0, cross.B) else (b0 * cross_group * T, b1 * cross_group * T)
                        K.attention_f32(q32[r0:r1], cross.k[i][b0:b1], cross.vt[i][b0:b1], o3[r0:r1], Bq=(r1 - r0) // T, H=H, Nq=T, Nk=cross.Te,
                                        kv_rows=cross.Tk_cap, kv_group=cross_group, kv_index=cross_index, group_start=cross_groups,
                                        max_group=cross_max_group, arith=1, kv16=True, planes=planes_h)
                else:
                    kv = cross.k[i]                                     # f32 [B, Te, 2C]: keys | values
                    K.attention_f32(q32, kv[..., :C], kv[..., C:], o3, Bq=rows, H=H, Nq=T, Nk=cross.Te, kv_rows=cross.Te,
                                    kv_group=cross_group, kv_index=cross_index, group_start=cross_groups, max_group=cross_max_group,
                                    arith=arith, planes=planes_h)
                K.gemm(o3, d["co_w3"], d["co_b"], out=tmp, resid=h32, split_k=True, a_planes=planes_h)
                K.layernorm(tmp, d["co_g"], d["co_bt"], eps, out16=h3, out32=h32, split3=True, planes=planes_h)
            elif cross is not None:
                K.gemm(h3, d["cq_w3"], d["cq_b"], heads=dict(q=q, T=T, H=H, part0=0, Tq_cap=T, q_scale=0.125), split_k=True)
                # (project_cross_kv(last_layer_vt=True) keeps the LAST layer's values in a V^T buffer of their own)
                last = i == len(p["layers"]) - 1 and cross.last_vt is not None
                K.attention(q, cross.k[i], cross.last_vt if last else cross.vt[i], o3, Bq=rows, H=H, Nq=T, Nk=cross.Te, Tq_cap=T,
                            Tk_cap=cross.Tk_cap, NP=cross.last_NP if last else cross.NP, kv_group=cross_group, kv_index=cross_index,
                            group_start=cross_groups, max_group=cross_max_group, kv_tiled=cross.tiled, split3=True)
                K.gemm(o3, d["co_w3"], d["co_b"], out=tmp, resid=h32, split_k=True)
                K.layernorm(tmp, d["co_g"], d["co_bt"], eps, out16=h3, out32=h32, split3=True, planes=planes_h)
            K.gemm(h3, d["i_w3"], d["i_b"], split3_out=inter3, act=K.ACT_GELU_ERF, split_k=True, split3_planes=planes, a_planes=planes_h)
            K.gemm(inter3, d["o_w3"], d["o_b"], out=tmp, resid=h32, split_k=True, a_planes=planes)
            K.layernorm(tmp, d["o_g"], d["o_bt"], eps, out16=h3, out32=h32, split3=True, planes=planes_h)
        return h32, h3

    def _text_fold_ok(self, cdt):
        C = self.config.hidden_size
        return (os.environ.get("VIDIL_FUSE_LN", "1") != "0" and C % 64 == 0 and C <= 1024
                and cdt in (torch.float16, torch.bfloat16))

    def _run_layers_fused(self, h32, h16, *, rows, T, self_k, self_vt, t_off, Tk_cap, NPs, causal, kv_len, cross, cross_index=None,
                          cross_group=1, cross_groups=None, cross_max_group=0, n_layers=None, self_done_first=False,
                          stop_after_self=False, state_in=None, arena: "BeamArena" = None, arena_slot_stride=1):
        """run_layers for encoder batches with cross-attention, WITHOUT LayerNorm launches between the GEMMs
        (models/med.py:236-239,306-317 are post-LN: h = LN(x + dense(.)) is the next dense's input AND the next residual).
        The stream is kept as the RAW sums u (f32 in h32's storage, a 16-bit copy in h16's) plus per-row (sum, sum of
        squares) partials written by the producing GEMM's epilogue; a consumer GEMM normalises its A rows algebraically
        (ln_fold: gamma-scaled weights, mean / rstd applied to the accumulators) and a residual GEMM normalises the
        residual rows on the way in (rln_gamma / rln_beta).  One LayerNorm launch at the end restores the (h32, h16)
        contract for the caller.  Same arithmetic as run_layers up to the rounding point of the GEMM operands (raw u
        instead of LN(u) is rounded to 16 bits).
        stop_after_self: run layer 0's self-attention block only and return the RAW state (h32, h16, partials, (gamma,
        beta)) — what encode_cls computes once per distinct text; state_in = (partials, (gamma, beta)): start from such
        a state (h32 / h16 hold raw sums), so the per-text front and the per-pair rest are the same arithmetic as one
        pass over the expanded batch.
        arena (round 6, the caption decoder: models/med.py:228-239,291-317,333-383 through the same folds): as run_layers — with
        T == 1 the new key / value of row r go to arena[position t_off][slot r] straight from the (LN-folded) Q|K|V GEMM's epilogue
        and attention follows the ancestry table; the prompt block (T > 1, t_off == 0) attends to itself and its K / V are also
        written to the arena at slots r * arena_slot_stride by a second, equally folded, K|V GEMM."""
        p = self.packed()
        fw = self._folded(p)
        cfg = self.config
        H, C = cfg.num_attention_heads, cfg.hidden_size
        eps = cfg.layer_norm_eps
        M = rows * T
        dev, cdt = h32.device, h16.dtype
        q = torch.empty((rows, H, T, 64), dtype=cdt, device=dev)
        o = torch.empty((M, C), dtype=cdt, device=dev)
        inter = torch.empty((M, cfg.intermediate_size), dtype=cdt, device=dev)
        stats = [torch.empty((M, C // 64, 2), dtype=torch.float32, device=dev) for _ in range(2)]
        cur = 0                      # stats[cur] describes the raw stream when `pend` is set
        pend = None                  # (gamma, beta) of the LayerNorm still owed to the stream; None: h32 / h16 are normalised
        if state_in is not None:
            stats[0], pend = state_in
        Nk = t_off + T

        def residual_gemm(a, w, b, ln_g, ln_b):
            """stream <- LN_pending(stream) + a·w^T + b, left RAW with fresh partials; (ln_g, ln_b) become pending."""
            nonlocal cur, pend
            kw = dict(out=h32, resid=h32, out16=h16, ln_stats_out=stats[cur ^ 1])
            if pend is not None:
                kw["rln"] = (pend[0], pend[1], eps, stats[cur])
            K.gemm(a, w, b, **kw)
            cur ^= 1
            pend = (ln_g, ln_b)

        def consumer(name, i, w, b, **kw):
            """a GEMM whose A operand is the stream: plain when it is normalised, LN-folded when a LayerNorm is pending."""
            if pend is None:
                return K.gemm(h16, w, b, **kw)
            wf, bf, cs = fw[i][name]
            return K.gemm(h16, wf, bf, ln=(cs, eps, stats[cur]), **kw)

        if arena is not None and T > 1 and t_off != 0:
            raise K.VidilHipError("run_layers: a multi-token block can only be appended to a beam arena at position 0")

        def arena_kv(i, d):
            """the prompt block's K | V once more, in the arena layout (a prompt is a few tokens): the K|V rows of the same
            (folded) weights — same values as the per-head K / V the block attends to"""
            kw = dict(arena=dict(k=arena.k[i], v=arena.v[i], T=T, H=H, part0=1, t_off=0, Tcap=arena.Tcap, arena_rows=arena.rows,
                                 slot_stride=arena_slot_stride))
            if pend is None:
                return K.gemm(h16, d["qkv_w"][C:], d["qkv_b"][C:], **kw)
            wf, bf, cs = fw[i]["qkv"]
            return K.gemm(h16, wf[C:], bf[C:], ln=(cs[C:], eps, stats[cur]), **kw)

        layers = p["layers"][:n_layers]
        for i, d in enumerate(layers):
            if self_done_first and i == 0:
                pass
            elif arena is not None and T == 1:
                consumer("qkv", i, d["qkv_w"], d["qkv_b"],
                         arena=dict(q=q, k=arena.k[i], v=arena.v[i], T=1, H=H, part0=0, t_off=t_off, Tcap=arena.Tcap,
                                    arena_rows=arena.rows, slot_stride=1, q_scale=0.125))
                K.beam_attention(q, arena.k[i], arena.v[i], arena.anc, o, rows=rows, H=H, n_keys=Nk)
                residual_gemm(o, d["ao_w"], d["ao_b"], d["ao_g"], d["ao_bt"])
            else:
                if arena is not None:
                    arena_kv(i, d)                     # (before the residual GEMM below replaces the stream it reads)
                consumer("qkv", i, d["qkv_w"], d["qkv_b"],
                         heads=dict(q=q, k=self_k[i], vt=self_vt[i], T=T, H=H, part0=0, t_off=t_off, Tq_cap=T, Tk_cap=Tk_cap,
                                    NP=NPs, q_scale=0.125))
                K.attention(q, self_k[i], self_vt[i], o, Bq=rows, H=H, Nq=T, Nk=Nk, Tq_cap=T, Tk_cap=Tk_cap, NP=NPs,
                            causal=causal, causal_off=t_off, kv_len=kv_len)
                residual_gemm(o, d["ao_w"], d["ao_b"], d["ao_g"], d["ao_bt"])
            if stop_after_self:
                return h32, h16, stats[cur], pend
            consumer("cq", i, d["cq_w"], d["cq_b"], heads=dict(q=q, T=T, H=H, part0=0, Tq_cap=T, q_scale=0.125))
            for b0, b1 in self._cross_blocks(cross, cross_index, cross_groups):
                r0, r1 = (0, rows) if (b0, b1) == (0, cross.B) else (b0 * cross_group, b1 * cross_group)
                K.attention(q[r0:r1], cross.k[i][b0:b1], cross.vt[i][b0:b1], o[r0 * T:r1 * T], Bq=r1 - r0, H=H, Nq=T, Nk=cross.Te,
                            Tq_cap=T, Tk_cap=cross.Tk_cap, NP=cross.NP, kv_group=cross_group, kv_index=cross_index,
                            group_start=cross_groups, max_group=cross_max_group, kv_tiled=cross.tiled)
            residual_gemm(o, d["co_w"], d["co_b"], d["co_g"], d["co_bt"])
            consumer("fc1", i, d["i_w"], d["i_b"], out=inter, act=K.ACT_GELU_ERF)
            residual_gemm(inter, d["o_w"], d["o_b"], d["o_g"], d["o_bt"])
        if pend is not None:         # hand the caller normalised (h32, h16) again
            K.layernorm(h32, pend[0], pend[1], eps, out16=h16, out32=h32)
        return h32, h16

    def embed(self, ids_i32, T, pos_off):
        """ids int32 [rows*T] -> (h32, h16) after the embedding LayerNorm (models/med.py:71-94)."""
        p = self.packed()
        C = self.config.hidden_size
        M = ids_i32.numel()
        dev = ids_i32.device
        cdt = p["layers"][0]["qkv_w"].dtype
        raw = torch.empty((M, C), dtype=torch.float32, device=dev)
        K.embed_tokens(ids_i32, p["word"], p["pos"], raw, T=T, pos_off=pos_off)
        h32 = torch.empty((M, C), dtype=torch.float32, device=dev)
        par = p["parity"]                 # parity precision mode: the 16-bit companion is [hi | lo | hi] rows
        h16 = torch.empty((M, 3 * C if par else C), dtype=cdt, device=dev)
        K.layernorm(raw, p["emb_g"], p["emb_b"], self.config.layer_norm_eps, out16=h16, out32=h32, split3=par)
        return h32, h16

    def encode(self, ids_i32, kv_len_i32, cross: CrossKV, cross_index=None, cross_groups=None, cross_max_group=0):
        """ITM encoder pass (models/blip_itm.py:51-56): ids int32 [P,T] right-padded, kv_len [P] = number of
        real tokens.  Pair p attends to image ``cross_index[p]``, or — image-major pair order — image j serves
        pairs cross_groups[j] .. cross_groups[j+1]-1.  Returns (h32, h16) [P*T, C]."""
        require_cuda(ids_i32, "BertModel.encode")
        P, T = ids_i32.shape
        H = self.config.num_attention_heads
        L = self.config.num_hidden_layers
        dev = ids_i32.device
        NPs = (T + 15) // 16 * 16
        h32, h16 = self.embed(ids_i32.reshape(-1), T, 0)
        cdt = h16.dtype
        sk = torch.empty((1, P, H, T, 64), dtype=cdt, device=dev).expand(L, -1, -1, -1, -1)
        sv = torch.empty((1, P, H, 64, NPs), dtype=cdt, device=dev).expand(L, -1, -1, -1, -1)
        # (one scratch K / V^T buffer is reused by every layer: the encoder keeps no cache)
        self.run_layers(h32, h16, rows=P, T=T, self_k=sk, self_vt=sv, t_off=0, Tk_cap=T, NPs=NPs, causal=False,
                        kv_len=kv_len_i32, cross=cross, cross_index=cross_index, cross_groups=cross_groups,
                        cross_max_group=cross_max_group)
        return h32, h16

    def encode_cls(self, ids_i32, kv_len_i32, cross: CrossKV, cross_index=None, cross_groups=None, cross_max_group=0,
                   pair_text=None):
        """encode() for consumers of the [CLS] position only (the ITM head, models/blip_itm.py:57): same arguments,
        returns (h32, h16) [P, C] — token 0 of every pair after the last layer.

        Layers 0..L-2 run on every token.  In the last layer only the keys and values of the self-attention depend on
        the other tokens, so it computes K|V for all P*T rows and everything else — the query, both attention
        outputs, the three dense+LayerNorm blocks and the feed-forward — for the P [CLS] rows alone (1/T of the
        layer's GEMM rows).  The arithmetic per [CLS] row is unchanged.  Needs ``cross.last_vt`` when the cross
        values are row-major (one query row per pair goes through the direct attention kernel).

        ``pair_text`` (int64 [P], device): ids / kv_len then describe U distinct TEXTS and pair p uses text
        pair_text[p] (CapFilt scores every caption of a video against each of its F frames).  Everything before
        the first cross-attention — the embedding and layer 0's self-attention block — is computed once per text
        and its rows are copied out to the pairs; same bits as running it per pair."""
        require_cuda(ids_i32, "BertModel.encode_cls")
        p = self.packed()
        cfg = self.config
        T = ids_i32.shape[1]
        H, C, L = cfg.num_attention_heads, cfg.hidden_size, cfg.num_hidden_layers
        eps = cfg.layer_norm_eps
        dev = ids_i32.device
        cdt = p["layers"][0]["qkv_w"].dtype
        NPs = (T + 15) // 16 * 16
        if cross is not None and cross.NP == 0 and cross.last_vt is None:
            raise K.VidilHipError("encode_cls: row-major cross values need project_cross_kv(last_layer_vt=True)")

        def scratch(rows):
            k = torch.empty((rows, H, T, 64), dtype=cdt, device=dev)
            v = torch.empty((rows, H, 64, NPs), dtype=cdt, device=dev)
            return k, v, k.unsqueeze(0).expand(L, -1, -1, -1, -1), v.unsqueeze(0).expand(L, -1, -1, -1, -1)

        h32, h16 = self.embed(ids_i32.reshape(-1), T, 0)
        shared = pair_text is not None and cross is not None and L > 1
        fold = self._text_fold_ok(cdt) and cross is not None and L > 1
        state = None
        if shared:
            U = ids_i32.shape[0]
            _, _, uk, uv = scratch(U)
            kw = dict(rows=U, T=T, self_k=uk, self_vt=uv, t_off=0, Tk_cap=T, NPs=NPs, causal=False, kv_len=kv_len_i32, cross=None,
                      n_layers=1, stop_after_self=True)
            if fold:     # the text-only front leaves the raw sum + partials, exactly as the expanded batch's layer 0 would
                h32, h16, st0, pend0 = self._run_layers_fused(h32, h16, **kw)
                state = (st0.view(U, T, -1).index_select(0, pair_text).view(-1, st0.shape[1], 2).contiguous(), pend0)
            else:
                self.run_layers(h32, h16, **kw)
            h32 = h32.view(U, T, C).index_select(0, pair_text).view(-1, C)
            h16 = h16.view(U, T, C).index_select(0, pair_text).view(-1, C)
            kv_len_i32 = kv_len_i32.index_select(0, pair_text).contiguous()
        elif pair_text is not None:
            h32 = h32.view(-1, T, C).index_select(0, pair_text).view(-1, C)
            h16 = h16.view(-1, T, C).index_select(0, pair_text).view(-1, C)
            kv_len_i32 = kv_len_i32.index_select(0, pair_text).contiguous()
        P = h32.shape[0] // T
        sk, sv, sk_l, sv_l = scratch(P)
        kw = dict(rows=P, T=T, self_k=sk_l, self_vt=sv_l, t_off=0, Tk_cap=T, NPs=NPs, causal=False, kv_len=kv_len_i32, cross=cross,
                  cross_index=cross_index, cross_groups=cross_groups, cross_max_group=cross_max_group, n_layers=L - 1,
                  self_done_first=shared)
        if state is not None:
            self._run_layers_fused(h32, h16, state_in=state, **kw)
        else:
            self.run_layers(h32, h16, **kw)
        d = p["layers"][L - 1]
        q1 = torch.empty((P, H, 1, 64), dtype=cdt, device=dev)
        o1 = torch.empty((P, C), dtype=cdt, device=dev)
        tmp = torch.empty((P, C), dtype=torch.float32, device=dev)
        c16 = torch.empty((P, C), dtype=cdt, device=dev)
        c32 = h32.view(P, T, C)[:, 0].contiguous()
        # self-attention: keys / values of every token, query of token 0 (A rows p*T of h16: strided operand)
        K.gemm(h16, d["qkv_w"][C:], d["qkv_b"][C:], heads=dict(k=sk, vt=sv, T=T, H=H, part0=1, t_off=0, Tk_cap=T, NP=NPs))
        K.gemm(h16.view(-1), d["qkv_w"][:C], d["qkv_b"][:C], M=P, lda=T * C,
               heads=dict(q=q1, T=1, H=H, part0=0, Tq_cap=1, q_scale=0.125))
        K.attention(q1, sk, sv, o1, Bq=P, H=H, Nq=1, Nk=T, Tq_cap=1, Tk_cap=T, NP=NPs, causal=False, causal_off=0,
                    kv_len=kv_len_i32)
        K.gemm(o1, d["ao_w"], d["ao_b"], out=tmp, resid=c32)
        K.layernorm(tmp, d["ao_g"], d["ao_bt"], eps, out16=c16, out32=c32)
        if cross is not None:
            vt, NP = (cross.vt[L - 1], cross.NP) if cross.last_vt is None else (cross.last_vt, cross.last_NP)
            K.gemm(c16, d["cq_w"], d["cq_b"], heads=dict(q=q1, T=1, H=H, part0=0, Tq_cap=1, q_scale=0.125))
            K.attention(q1, cross.k[L - 1], vt, o1, Bq=P, H=H, Nq=1, Nk=cross.Te, Tq_cap=1, Tk_cap=cross.Te, NP=NP,
                        kv_index=cross_index, group_start=cross_groups, max_group=cross_max_group)
            K.gemm(o1, d["co_w"], d["co_b"], out=tmp, resid=c32)
            K.layernorm(tmp, d["co_g"], d["co_bt"], eps, out16=c16, out32=c32)
        inter = K.gemm(c16, d["i_w"], d["i_b"], act=K.ACT_GELU_ERF)
        K.gemm(inter, d["o_w"], d["o_b"], out=tmp, resid=c32)
        K.layernorm(tmp, d["o_g"], d["o_bt"], eps, out16=c16, out32=c32)
        return c32, c16

    def forward(self, *a, **k):
        raise NotImplementedError("use BertModel.encode / BLIP_ITM on the hot path")


class _LMTransform(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _LMPredictions(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.transform = _LMTransform(cfg)
        self.decoder = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(cfg.vocab_size))
        self.decoder.bias = self.bias  # same aliasing as models/med.py:527-530


class _LMHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.predictions = _LMPredictions(cfg)


class BertLMHeadModel(PackedCache, nn.Module):
    """Caption decoder: ``bert`` trunk + ``cls`` LM head (models/med.py:811-955).

    ``precise_head`` (default: $VIDIL_PRECISE_LM_HEAD == "1"): run the two GEMMs of the LM head with
    error-compensated operands — activations and weights split into hi + lo 16-bit parts, the three significant
    cross terms accumulated by ONE GEMM with K tripled — so that the head adds ~2^-20 instead of ~2^-11 (f16) /
    2^-8 (bf16) relative rounding to the logits.  It is what the "caption logits within 1e-3" budget can afford
    to spend on precision: the head is 19 % of a decode step's FLOPs (x3 here), the trunk's operand rounding is
    untouched (DESIGN.md §4 has the measured split)."""

    def __init__(self, config):
        super().__init__()
        import os

        self.precise_head = os.environ.get("VIDIL_PRECISE_LM_HEAD", "0") == "1"
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = _LMHead(config)
        self.cls.apply(lambda m: _init_bert(m, config.initializer_range))

    def _pack(self):
        pr = self.cls.predictions
        c = self.cdt
        p = dict(t_w=w16(pr.transform.dense.weight, dtype=c), t_b=v32(pr.transform.dense.bias),
                 t_g=v32(pr.transform.LayerNorm.weight), t_bt=v32(pr.transform.LayerNorm.bias),
                 dec_w=w16(pr.decoder.weight, dtype=c), dec_b=v32(pr.bias), precise=self.precise_head or self.parity)
        if p["precise"]:
            p.update(t_w3=w3(pr.transform.dense.weight, dtype=c), dec_w3=w3(pr.decoder.weight, dtype=c))
        return p

    def pack_flags(self):
        return (self.precise_head, self.parity)

    def lm_logits(self, h16, rows, T, out=None, h32=None):
        """LM head (models/med.py:501-545) on the LAST token of each of ``rows`` sequences of length T:
        dense -> erf-GELU -> LayerNorm -> decoder(+bias).  h16: T16 [rows*T, C]; h32: the same hidden states in f32
        (used by the precise head).  Returns f32 [rows, V].
        (The reference computes logits for all T positions and HF generate() keeps only the last.)"""
        p = self.packed()
        cfg = self.config
        C = cfg.hidden_size
        dev = h16.device
        cdt = h16.dtype
        if p["precise"]:
            if h32 is None:
                raise K.VidilHipError("lm_logits: the error-compensated head (precise_head / parity mode) needs the f32 hidden states")
            last32 = h32.view(rows, T, C)[:, T - 1].contiguous()
            a3 = K.split3(last32, torch.empty((rows, 3 * C), dtype=cdt, device=dev))
            t32 = torch.empty((rows, C), dtype=torch.float32, device=dev)
            K.gemm(a3, p["t_w3"], p["t_b"], out=t32, act=K.ACT_GELU_ERF, split_k=True)
            tn32 = torch.empty((rows, C), dtype=torch.float32, device=dev)
            K.layernorm(t32, p["t_g"], p["t_bt"], cfg.layer_norm_eps, out32=tn32)
            K.split3(tn32, a3)
            if out is None:
                out = torch.empty((rows, cfg.vocab_size), dtype=torch.float32, device=dev)
            K.gemm(a3, p["dec_w3"], p["dec_b"], out=out, split_k=True)
            return out
        last = h16.view(-1)[(T - 1) * C:]  # row r of the strided view = token T-1 of sequence r
        t32 = torch.empty((rows, C), dtype=torch.float32, device=dev)
        K.gemm(last, p["t_w"], p["t_b"], out=t32, act=K.ACT_GELU_ERF, M=rows, lda=T * C)
        t16 = torch.empty((rows, C), dtype=cdt, device=dev)
        K.layernorm(t32, p["t_g"], p["t_bt"], cfg.layer_norm_eps, out16=t16)
        if out is None:
            out = torch.empty((rows, cfg.vocab_size), dtype=torch.float32, device=dev)
        K.gemm(t16, p["dec_w"], p["dec_b"], out=out)
        return out

    def forward(self, *a, **k):
        raise NotImplementedError("use vidil_amd.blip.BLIP_Decoder.generate on the hot path")
