"""MI355X-native mirror of the reference's `model/caption_decoder.py` hot path (reference
model/caption_decoder.py:526-613 `CaptionDecoder`, :316-423 `Mesh_TransformerDecoderLayer`, :272-314
`PositionalEncoding`): same constructor `CaptionDecoder(args)` (args.vocab_size, embed_dim, n_head, n_layer,
dropout), same `forward(memory, encoded_captions, caption_lengths) -> (pred, caps_sorted, decode_lengths, sort_ind)`,
same state-dict keys (`vocab_embedding.weight`, `transformer.layers.{i}.self_attn.in_proj_weight`, ...,
`position_encoding.pe`, `wdc.*`) INCLUDING the modules the reference constructs but never runs (self_attn2,
multihead_attn, multihead_attn3, linear1/2, norm3, fc_alpha1-3, embedding_1D) so checkpoints strict-load both ways.

The nn children are parameter holders; the computation is HIP kernels (csrc/caption_ops.hip + the wide GEMM of
csrc/pw_wide.hip) behind one autograd function:

    x = drop_0.1(embed(tokens) + pe)                                   # reference quirk: 0.1 whatever --dropout says
    per layer:  x = LN1(x + drop(SelfAttn(x, causal)));  x = LN2(x + drop(CrossAttn2(x, memory)))   # no FFN in forward
    pred = wdc(drop(x))  ->  (B, L, vocab), sorted by caption length

Dropout masks come from a counter-based generator (seed drawn from torch's CPU generator per call), so train-mode
results are reproducible under `torch.manual_seed` but are NOT bit-comparable with torch's CPU dropout stream; parity
tests run with dropout 0 (tests/test_cc_gpu.py).
"""
import copy
import math

import torch
import torch.nn as nn

from .. import ops
from ..ops import cpad
from .utils import weight_init


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1))
        self.embedding_1D = nn.Embedding(52, int(d_model))     # constructed, never used (reference :299)

    def forward(self, x):
        """(L, B, D) embeddings + positions (reference :301-314); used by the reference's evaluation loop, which calls
        the sub-modules one by one (scripts/train_CC.py:260-261).  The training path fuses this into c3d_cap_embed_fwd."""
        return self.dropout(x + self.pe[:x.size(0), :])


class Mesh_TransformerDecoderLayer(nn.Module):
    """Parameter holder with the reference layer's attribute names (reference model/caption_decoder.py:316-383)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, layer_norm_eps=1e-5):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.self_attn2 = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(int(d_model), nhead, dropout=dropout)
        self.multihead_attn2 = nn.MultiheadAttention(d_model, int(nhead), dropout=dropout)
        self.multihead_attn3 = nn.MultiheadAttention(int(d_model), int(nhead), dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.dropout4, self.dropout5 = nn.Dropout(dropout), nn.Dropout(dropout)
        self.activation, self.activation2 = nn.ReLU(), nn.Softmax(dim=-1)
        self.fc_alpha1 = nn.Linear(d_model + d_model, d_model)
        self.fc_alpha2 = nn.Linear(d_model + d_model, d_model)
        self.fc_alpha3 = nn.Linear(d_model + d_model, d_model)
        for fc in (self.fc_alpha1, self.fc_alpha2, self.fc_alpha3):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)
        weight_init(self)


class _Layers(nn.Module):
    """Key-compatible stand-in for `nn.TransformerDecoder(layer, n)` (`.layers` = n deep copies, no final norm)."""

    def __init__(self, layer, n):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
        self.norm = None
        self.n_head, self.act_dtype = layer.self_attn.num_heads, torch.float32

    def forward(self, tgt, memory, tgt_mask=None, **_unused):
        """Inference-only `nn.TransformerDecoder.forward(tgt, memory, tgt_mask=causal)` on the HIP kernels: what the
        reference's beam search calls at every step (scripts/train_CC.py:264).  tgt (L, B, D), memory (S, B, D) ->
        (L, B, D).  Training goes through `CaptionDecoder.forward` (one fused autograd node)."""
        if torch.is_grad_enabled() and (tgt.requires_grad or memory.requires_grad or self.training):
            raise NotImplementedError("_Layers.forward is the no-grad evaluation path; train through CaptionDecoder.forward")
        L, B, D = tgt.shape
        if tgt_mask is not None:
            ref = torch.triu(torch.ones(L, L, device=tgt_mask.device), diagonal=1) > 0
            if tgt_mask.shape != (L, L) or not bool(((tgt_mask == float("-inf")) == ref).all()):
                raise NotImplementedError("only the reference's causal mask is supported")
        ops.require_gpu(tgt, "caption decoder input")
        act = self.act_dtype
        x = tgt.detach().to(act).contiguous().view(L * B, D)
        kv = project_memory(self, memory)
        h = _infer_layers(self, x, kv, memory.shape[0], B, L, causal=tgt_mask is not None)
        return h.view(L, B, -1)[:, :, :D].to(tgt.dtype)


def project_memory(stack, memory):
    """Per layer, the key/value projection of the encoder memory (S, B, D) -> [S*B][2D] rows: constant over the steps
    of a beam search, so it is computed once per image pair."""
    S, B, D = memory.shape
    act = stack.act_dtype
    dt = ops.dt_code(act)
    mem = memory.detach().to(act).contiguous().view(S * B, D)
    out = []
    for layer in stack.layers:
        ca = layer.multihead_attn2
        kv2 = torch.empty((S * B, 2 * D), dtype=act, device=mem.device)
        ops.linear_fwd(mem, ca.in_proj_weight[D:], ca.in_proj_bias[D:], kv2, S * B, D, 2 * D, dt)
        out.append(kv2)
    return out


def _infer_layers(stack, x, kv, S, B, L, causal=True):
    """Evaluation-mode layer stack (no dropout, nothing saved): x rows [L*B][D] sequence-first -> rows [L*B][D]."""
    D = x.shape[1]
    H = stack.n_head
    hd = D // H
    act, dev = x.dtype, x.device
    dt = ops.dt_code(act)
    if cpad(D) != D:
        raise NotImplementedError("embed_dim must be a multiple of 8")
    R, scale = L * B, 1.0 / math.sqrt(hd)
    new = lambda *s: torch.empty(s, dtype=act, device=dev)  # noqa: E731
    P1 = torch.empty((H * B, L, L), dtype=torch.float32, device=dev)
    P2 = torch.empty((H * B, L, S), dtype=torch.float32, device=dev)
    mr = torch.empty((R, 2), dtype=torch.float32, device=dev)
    qkv, o, a, q2 = new(R, 3 * D), new(R, D), new(R, D), new(R, D)
    for layer, kv2 in zip(stack.layers, kv):
        sa, ca = layer.self_attn, layer.multihead_attn2
        ops.linear_fwd(x, sa.in_proj_weight, sa.in_proj_bias, qkv, R, D, 3 * D, dt)
        ops.cap_attn_fwd(qkv, qkv, qkv, 3 * D, 3 * D, 3 * D, o, D, P1, B, H, L, L, hd, scale, causal, 0.0, 0, dt,
                         q_off=0, k_off=D, v_off=2 * D)
        ops.linear_fwd(o, sa.out_proj.weight, sa.out_proj.bias, a, R, D, D, dt)
        x1 = new(R, D)
        ops.cap_layernorm_fwd(x, a, layer.norm1, x1, mr, R, D, dt)
        ops.linear_fwd(x1, ca.in_proj_weight[:D], ca.in_proj_bias[:D], q2, R, D, D, dt)
        ops.cap_attn_fwd(q2, kv2, kv2, D, 2 * D, 2 * D, o, D, P2, B, H, L, S, hd, scale, False, 0.0, 0, dt, k_off=0, v_off=D)
        ops.linear_fwd(o, ca.out_proj.weight, ca.out_proj.bias, a, R, D, D, dt)
        x2 = new(R, D)
        ops.cap_layernorm_fwd(x1, a, layer.norm2, x2, mr, R, D, dt)
        x = x2
    return x


def _mha_used_params(m):
    return [m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias]


class _CaptionFn(torch.autograd.Function):
    """Whole decoder forward / backward (embedding .. vocabulary logits) as one autograd node."""

    @staticmethod
    def forward(ctx, memory, dec, caps, seed, *params):
        ops.require_gpu(memory, "caption decoder memory")
        S, B, D = memory.shape
        L = caps.shape[1]
        H, hd = dec.n_head, D // dec.n_head
        act = dec.act_dtype
        dt, dev = ops.dt_code(act), memory.device
        train = dec.training
        p_attn = dec.dropout_p if train else 0.0
        p_pos = dec.position_encoding.dropout.p if train else 0.0
        p_out = dec.dropout_layer.p if train else 0.0
        Dp, V = cpad(D), dec.vocab_size
        R = L * B
        scale = 1.0 / math.sqrt(hd)
        mem = memory.detach().to(act).contiguous().view(S * B, D)
        if Dp != D:
            raise NotImplementedError("embed_dim must be a multiple of 8")
        tok = caps.detach().contiguous()
        x = torch.empty((R, Dp), dtype=act, device=dev)
        pe = dec.position_encoding.pe.view(-1, D)
        ops.cap_embed_fwd(tok, dec.vocab_embedding.weight, pe, x, B, L, D, V, p_pos, seed, dt)
        saved = []
        for li, layer in enumerate(dec.transformer.layers):
            sa, ca = layer.self_attn, layer.multihead_attn2
            sd = seed + 1000 * (li + 1)
            # ---- causal self-attention
            qkv = torch.empty((R, 3 * D), dtype=act, device=dev)
            ops.linear_fwd(x, sa.in_proj_weight, sa.in_proj_bias, qkv, R, D, 3 * D, dt)
            P1 = torch.empty((H * B, L, L), dtype=torch.float32, device=dev)
            o1 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.cap_attn_fwd(qkv, qkv, qkv, 3 * D, 3 * D, 3 * D, o1, Dp, P1, B, H, L, L, hd, scale, True, p_attn, sd + 1, dt,
                             q_off=0, k_off=D, v_off=2 * D)
            a1 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_fwd(o1, sa.out_proj.weight, sa.out_proj.bias, a1, R, D, D, dt)
            if p_attn > 0:
                ops.cap_dropout(a1, a1, R, D, p_attn, sd + 2, dt)           # dropout1
            x1 = torch.empty((R, Dp), dtype=act, device=dev)
            mr1 = torch.empty((R, 2), dtype=torch.float32, device=dev)
            ops.cap_layernorm_fwd(x, a1, layer.norm1, x1, mr1, R, D, dt)
            # ---- cross-attention over the encoder memory (multihead_attn2)
            q2 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_fwd(x1, ca.in_proj_weight[:D], ca.in_proj_bias[:D], q2, R, D, D, dt)
            kv2 = torch.empty((S * B, 2 * D), dtype=act, device=dev)
            ops.linear_fwd(mem, ca.in_proj_weight[D:], ca.in_proj_bias[D:], kv2, S * B, D, 2 * D, dt)
            P2 = torch.empty((H * B, L, S), dtype=torch.float32, device=dev)
            o2 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.cap_attn_fwd(q2, kv2, kv2, Dp, 2 * D, 2 * D, o2, Dp, P2, B, H, L, S, hd, scale, False, p_attn, sd + 3, dt,
                             k_off=0, v_off=D)
            a2 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_fwd(o2, ca.out_proj.weight, ca.out_proj.bias, a2, R, D, D, dt)
            if p_attn > 0:
                ops.cap_dropout(a2, a2, R, D, p_attn, sd + 4, dt)           # dropout3
            x2 = torch.empty((R, Dp), dtype=act, device=dev)
            mr2 = torch.empty((R, 2), dtype=torch.float32, device=dev)
            ops.cap_layernorm_fwd(x1, a2, layer.norm2, x2, mr2, R, D, dt)
            saved.append((x, qkv, P1, o1, a1, mr1, x1, q2, kv2, P2, o2, a2, mr2))
            x = x2
        xd = x
        if p_out > 0:
            xd = torch.empty_like(x)
            ops.cap_dropout(x, xd, R, D, p_out, seed + 7, dt)               # dropout_layer
        Vp = cpad(V)
        logits = torch.empty((R, Vp), dtype=act, device=dev)
        ops.linear_fwd(xd, dec.wdc.weight, dec.wdc.bias, logits, R, D, V, dt)
        ctx.dec, ctx.saved, ctx.tail = dec, saved, (xd, tok, mem)
        ctx.meta = (S, B, D, L, H, hd, act, p_attn, p_pos, p_out, seed, scale, memory.dtype)
        return logits          # sequence-first rows [L*B][Vp]

    @staticmethod
    def backward(ctx, dlogits):
        dec, saved = ctx.dec, ctx.saved
        xd, tok, mem = ctx.tail
        S, B, D, L, H, hd, act, p_attn, p_pos, p_out, seed, scale, mem_dtype = ctx.meta
        dt, dev = ops.dt_code(act), dlogits.device
        Dp, V, R = cpad(D), dec.vocab_size, L * B
        dl = dlogits.to(act).contiguous()
        dx = torch.empty((R, Dp), dtype=act, device=dev)
        ops.linear_bwd(xd, dec.wdc.weight, dl, dx, ops.grad_of(dec.wdc.weight), ops.grad_of(dec.wdc.bias), R, D, V, dt)
        if p_out > 0:
            ops.cap_dropout(dx, dx, R, D, p_out, seed + 7, dt)
        dmem = torch.zeros((S * B, Dp), dtype=act, device=dev)
        first_mem = True
        for li in range(len(saved) - 1, -1, -1):
            layer = dec.transformer.layers[li]
            sa, ca = layer.self_attn, layer.multihead_attn2
            sd = seed + 1000 * (li + 1)
            x0, qkv, P1, o1, a1, mr1, x1, q2, kv2, P2, o2, a2, mr2 = saved[li]
            # ---- x2 = LN2(x1 + a2)
            d12 = torch.empty((R, Dp), dtype=act, device=dev)        # gradient of x1 (residual) == gradient of a2
            ops.cap_layernorm_bwd(x1, a2, dx, layer.norm2, mr2, d12, R, D, dt)
            da2 = d12
            if p_attn > 0:
                da2 = torch.empty_like(d12)
                ops.cap_dropout(d12, da2, R, D, p_attn, sd + 4, dt)
            do2 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_bwd(o2, ca.out_proj.weight, da2, do2, ops.grad_of(ca.out_proj.weight), ops.grad_of(ca.out_proj.bias), R, D, D, dt)
            dq2 = torch.empty((R, Dp), dtype=act, device=dev)
            dkv2 = torch.empty((S * B, 2 * D), dtype=act, device=dev)
            ops.cap_attn_bwd(q2, kv2, kv2, Dp, 2 * D, 2 * D, do2, Dp, P2, dq2, dkv2, dkv2, Dp, 2 * D, 2 * D, B, H, L, S, hd, scale,
                             p_attn, sd + 3, dt, k_off=0, v_off=D, dk_off=0, dv_off=D)
            gw, gb = ops.grad_of(ca.in_proj_weight), ops.grad_of(ca.in_proj_bias)
            # memory projection: d mem += dkv2 @ W_kv ; parameter gradients into the [D:] rows
            ops.linear_bwd(mem, ca.in_proj_weight[D:], dkv2, dmem, gw[D:], gb[D:], S * B, D, 2 * D, dt,
                           accumulate_dx=None if first_mem else dmem)
            first_mem = False
            # query projection: d x1 = d12 (residual) + dq2 @ W_q
            dx1 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_bwd(x1, ca.in_proj_weight[:D], dq2, dx1, gw[:D], gb[:D], R, D, D, dt, accumulate_dx=d12)
            # ---- x1 = LN1(x0 + a1)
            d01 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.cap_layernorm_bwd(x0, a1, dx1, layer.norm1, mr1, d01, R, D, dt)
            da1 = d01
            if p_attn > 0:
                da1 = torch.empty_like(d01)
                ops.cap_dropout(d01, da1, R, D, p_attn, sd + 2, dt)
            do1 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_bwd(o1, sa.out_proj.weight, da1, do1, ops.grad_of(sa.out_proj.weight), ops.grad_of(sa.out_proj.bias), R, D, D, dt)
            dqkv = torch.empty((R, 3 * D), dtype=act, device=dev)
            ops.cap_attn_bwd(qkv, qkv, qkv, 3 * D, 3 * D, 3 * D, do1, Dp, P1, dqkv, dqkv, dqkv, 3 * D, 3 * D, 3 * D, B, H, L, L, hd,
                             scale, p_attn, sd + 1, dt, q_off=0, k_off=D, v_off=2 * D, dq_off=0, dk_off=D, dv_off=2 * D)
            dx0 = torch.empty((R, Dp), dtype=act, device=dev)
            ops.linear_bwd(x0, sa.in_proj_weight, dqkv, dx0, ops.grad_of(sa.in_proj_weight), ops.grad_of(sa.in_proj_bias), R, D, 3 * D,
                           dt, accumulate_dx=d01)
            dx = dx0
        ops.cap_embed_bwd(tok, dx, ops.grad_of(dec.vocab_embedding.weight), B, L, D, V, p_pos, seed, dt)
        gmem = dmem.view(S, B, Dp)[:, :, :D].to(mem_dtype) if ctx.needs_input_grad[0] else None
        return (gmem, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class CaptionDecoder(nn.Module):
    """Caption decoder for caption generation (reference model/caption_decoder.py:526-613)."""

    def __init__(self, args):
        super().__init__()
        print(f"decoder_n_layers={args.n_layer}")
        self.vocab_size, self.n_head, self.dropout_p = args.vocab_size, args.n_head, float(args.dropout)
        self.act_dtype = getattr(args, "act_dtype", torch.float32)
        self.vocab_embedding = nn.Embedding(args.vocab_size, args.embed_dim)
        layer = Mesh_TransformerDecoderLayer(args.embed_dim, args.n_head, dim_feedforward=args.embed_dim * 4,
                                             dropout=args.dropout)
        self.transformer = _Layers(layer, args.n_layer)
        self.transformer.act_dtype = self.act_dtype
        self.position_encoding = PositionalEncoding(args.embed_dim)
        self.wdc = nn.Linear(args.embed_dim, args.vocab_size)
        self.dropout_layer = nn.Dropout(p=args.dropout)
        self.init_weights()

    def init_weights(self):
        self.vocab_embedding.weight.data.uniform_(-0.1, 0.1)
        self.wdc.bias.data.fill_(0)
        self.wdc.weight.data.uniform_(-0.1, 0.1)

    def used_parameters(self):
        """The parameters the forward pass touches (the others never receive a gradient, as in the reference)."""
        ps = [self.vocab_embedding.weight, self.wdc.weight, self.wdc.bias]
        for layer in self.transformer.layers:
            ps += _mha_used_params(layer.self_attn) + _mha_used_params(layer.multihead_attn2)
            ps += [layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias]
        return ps

    def logits_seq_first(self, memory, encoded_captions):
        """Vocabulary logits as sequence-first rows [L*B][round_up(V,8)] (what the fused packed cross-entropy reads)."""
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if self.training else 0
        return _CaptionFn.apply(memory, self, encoded_captions, seed, *self.used_parameters())

    def forward(self, memory, encoded_captions, caption_lengths):
        B, L = encoded_captions.shape
        V = self.vocab_size
        logits = self.logits_seq_first(memory, encoded_captions)
        pred = logits.view(L, B, -1)[:, :, :V].permute(1, 0, 2).float()
        caption_lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True)
        encoded_captions = encoded_captions[sort_ind]
        pred = pred[sort_ind]
        decode_lengths = (caption_lengths - 1).tolist()
        return pred, encoded_captions, decode_lengths, sort_ind


    @torch.no_grad()
    def beam_search(self, encoder_out, start_id, end_id, beam_size, max_len=52):
        """Caption one image pair: the beam search of the reference's `evaluate()` (scripts/train_CC.py:214-330).
        encoder_out (S, 1, D) = rearrange(encoder(..., output_final=True), 'b c h w -> (h w) b c').
        Returns (best_seq or None, complete_seqs, complete_seqs_scores) -- None when no beam ever emits <end>, in
        which case the reference records no caption for the pair (:326-328).

        Same hypotheses as the reference, less work per step: the causal mask makes position step-1 independent of the
        (all-<pad>) positions after it, so only the first `step` tokens are decoded instead of the 52-token window
        (26x fewer rows on average); the vocabulary projection runs on the `s` rows of position step-1 only; the
        memory key/value projections are computed once per pair (all beams attend to the same memory, so shrinking the
        beam only drops columns).  Embedding, attention, LayerNorm and the projections are the HIP kernels of the
        training path; log-softmax / top-k over [s, vocab] are torch device ops (bookkeeping)."""
        was_training = self.training
        self.eval()
        try:
            return self._beam_search(encoder_out, start_id, end_id, beam_size, max_len)
        finally:
            self.train(was_training)

    def _beam_search(self, encoder_out, start_id, end_id, beam_size, max_len):
        ops.require_gpu(encoder_out, "caption decoder memory")
        S, one, D = encoder_out.shape
        if one != 1:
            raise ValueError("beam_search captions one image pair: encoder_out must be (S, 1, D)")
        k, V, dev, act = beam_size, self.vocab_size, encoder_out.device, self.act_dtype
        dt = ops.dt_code(act)
        kv1 = project_memory(self.transformer, encoder_out)                   # per layer [S][2D]
        kv_of = {}

        def kv_for(s):                                                        # [S*s][2D], rows (position, beam)
            if s not in kv_of:
                kv_of[s] = [t.view(S, 1, 2 * D).expand(S, s, 2 * D).contiguous().view(S * s, 2 * D) for t in kv1]
            return kv_of[s]

        pe = self.position_encoding.pe.view(-1, D)
        words = torch.zeros((k, max_len), dtype=torch.int64, device=dev)
        words[:, 0] = start_id
        seqs = torch.full((k, 1), start_id, dtype=torch.int64, device=dev)
        top_k_scores = torch.zeros((k, 1), dtype=torch.float32, device=dev)
        complete_seqs, complete_scores = [], []
        Vp = cpad(V)
        step = 1
        while True:
            s, L = words.shape[0], step
            tok = words[:, :L].contiguous()
            x = torch.empty((L * s, D), dtype=act, device=dev)
            ops.cap_embed_fwd(tok, self.vocab_embedding.weight, pe, x, s, L, D, V, 0.0, 0, dt)
            h = _infer_layers(self.transformer, x, kv_for(s), S, s, L, causal=True)
            logits = torch.empty((s, Vp), dtype=act, device=dev)
            ops.linear_fwd(h[(L - 1) * s:], self.wdc.weight, self.wdc.bias, logits, s, D, V, dt)
            scores = torch.log_softmax(logits[:, :V].float(), dim=1)
            scores = top_k_scores.expand_as(scores) + scores
            if step == 1:
                top_k_scores, top_k_words = scores[0].topk(k, 0, True, True)
            else:
                top_k_scores, top_k_words = scores.reshape(-1).topk(k, 0, True, True)
            prev_word_inds = top_k_words // V
            next_word_inds = top_k_words % V
            seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], dim=1)
            done = (next_word_inds == end_id)
            done_host = done.tolist()                                         # the one host sync per step
            if any(done_host):
                complete_seqs.extend(seqs[done].tolist())
                complete_scores.extend(top_k_scores[done].tolist())
            k -= sum(done_host)
            if k == 0:
                break
            keep = ~done
            seqs = seqs[keep]
            top_k_scores = top_k_scores[keep].unsqueeze(1)
            words = words[:k].clone()
            words[:, :step + 1] = seqs
            if step > 50:
                break
            step += 1
        if not complete_scores:
            return None, complete_seqs, complete_scores
        best = complete_scores.index(max(complete_scores))
        return complete_seqs[best], complete_seqs, complete_scores


class _PackedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_sf, caps, declen, ignore_index, V):
        ops.require_gpu(logits_sf, "caption logits")
        B, L = caps.shape
        dt = ops.dt_code(logits_sf.dtype)
        lg = logits_sf.detach().contiguous()
        acc = torch.empty(3, dtype=torch.float64, device=lg.device)
        lse = torch.empty(L * B, dtype=torch.float32, device=lg.device)
        loss = torch.empty(1, dtype=torch.float32, device=lg.device)
        ops.cap_ce_fwd(lg, caps, declen, acc, lse, loss, B, L, V, ignore_index, dt)
        ctx.saved, ctx.meta = (lg, caps, declen, acc, lse), (B, L, V, ignore_index, dt)
        ctx.mark_non_differentiable(acc)
        return loss[0], acc

    @staticmethod
    def backward(ctx, dloss, _dacc=None):
        lg, caps, declen, acc, lse = ctx.saved
        B, L, V, ignore_index, dt = ctx.meta
        d = torch.empty_like(lg)
        ops.cap_ce_bwd(lg, caps, declen, acc, lse, dloss.detach().reshape(1).contiguous().float(), d, B, L, V, ignore_index, dt)
        return d, None, None, None, None


def packed_cross_entropy(logits_seq_first, caps, caplens, vocab_size, ignore_index=0, return_stats=False):
    """`CrossEntropyLoss(ignore_index)(pack_padded_sequence(scores, decode_lengths).data, pack_padded_sequence(
    caps_sorted[:, 1:], decode_lengths).data)` of reference scripts/train_CC.py:124-132 as one fused pass: the mean
    over the decoded steps does not depend on the packing order, so neither the sort nor the gather is materialised.
    logits_seq_first: `CaptionDecoder.logits_seq_first(...)`; caps int64 [B, L]; caplens int64 [B, 1].
    `return_stats`: also return the device tensor f64 [3] = (sum nll, decoded steps, top-1 hits) -- the inputs of the
    reference's `caption_accuracy(scores, targets, 1)` and `losses.update(loss, sum(decode_lengths))` without a sync."""
    declen = (caplens.reshape(-1) - 1).contiguous()
    loss, acc = _PackedCEFn.apply(logits_seq_first, caps.contiguous(), declen, ignore_index, vocab_size)
    return (loss, acc) if return_stats else loss
