"""ORACLE -- test infrastructure only.

CPU restatement of the change-captioning path (BASELINE.json configs[4]):

  * `CaptionDecoder` (reference model/caption_decoder.py:526-613) with its `Mesh_TransformerDecoderLayer`
    (:316-423) and `PositionalEncoding` (:272-314): token embedding + sinusoidal positions, n_layer x
    { x = LN1(x + SelfAttn(x, causal mask)); x = LN2(x + CrossAttn2(x, memory)) } -- the layer constructs, but its
    forward never runs, self_attn2 / multihead_attn / multihead_attn3 / linear1 / linear2 / norm3 / fc_alpha1-3
    (they exist here too so that the state-dict keys are the reference's) -- vocabulary projection `wdc`, caption
    sort by length.
  * the training-step arithmetic of reference scripts/train_CC.py:118-147: encoder(output_final=True) ->
    rearrange 'b c h w -> (h w) b c' -> decoder -> pack_padded_sequence -> CrossEntropyLoss(ignore_index=0) ->
    backward -> clip_gradient -> two Adam optimisers (weight_decay 1e-5) with StepLR(step 900, gamma 1).

Two deviations from the reference's TEXT, both forced by running on torch 2.10 / CPU and both value-preserving
(SURVEY.md 8(c) "CC caveat"): the layers are applied in a Python loop instead of through `nn.TransformerDecoder`
(torch 2.10 passes `tgt_is_causal=` which the reference layer's forward does not accept; no final norm is
configured, so the loop IS the container's arithmetic), and the causal mask stays on the input's device instead of
the hard-coded `mask.cuda()`.

Pinned by: `tests/test_cpu.py::test_cc_oracle_equals_imported_reference` (state-dict keys, and bit-identical
logits / loss / every gradient against the REAL reference modules driven through the same per-layer loop) and the
fixture `tests/golden/cc_s256_b2.npz` written by `oracle/gen_golden.py::run_cc`."""
import copy
import math

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence

from .model import weight_init


class PositionalEncoding(nn.Module):
    """reference model/caption_decoder.py:272-314 (`embedding_1D` is constructed, never used)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1))
        self.embedding_1D = nn.Embedding(52, int(d_model))

    def forward(self, x):
        return self.dropout(x + self.pe[:x.size(0), :])


class MeshTransformerDecoderLayer(nn.Module):
    """reference model/caption_decoder.py:316-423: post-LN, causal self-attention then cross-attention through
    `multihead_attn2`; NO feed-forward block in forward."""

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
        for fc in (self.fc_alpha1, self.fc_alpha2, self.fc_alpha3):   # init_weights (:373-383)
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)
        weight_init(self)

    def forward(self, tgt, memory, tgt_mask=None):
        sa = self.self_attn(tgt, tgt, tgt, attn_mask=tgt_mask, key_padding_mask=None, need_weights=False)[0]
        x = self.norm1(tgt + self.dropout1(sa))
        enc, _ = self.multihead_attn2(x, memory, memory, attn_mask=None, key_padding_mask=None, need_weights=True)
        return self.norm2(x + self.dropout3(enc))


class _Layers(nn.Module):
    """Key-compatible stand-in for `nn.TransformerDecoder(layer, n)`: `.layers` = n deep copies, `.norm` = None."""

    def __init__(self, layer, n):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
        self.norm = None


def causal_mask(n, device=None):
    """reference model/caption_decoder.py:591-592."""
    mask = (torch.triu(torch.ones(n, n, device=device)) == 1).transpose(0, 1)
    return mask.float().masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, float(0.0))


class CaptionDecoder(nn.Module):
    """reference model/caption_decoder.py:526-613; args: vocab_size, embed_dim, n_head, n_layer, dropout."""

    def __init__(self, args):
        super().__init__()
        self.vocab_embedding = nn.Embedding(args.vocab_size, args.embed_dim)
        layer = MeshTransformerDecoderLayer(args.embed_dim, args.n_head, dim_feedforward=args.embed_dim * 4,
                                            dropout=args.dropout)
        self.transformer = _Layers(layer, args.n_layer)
        self.position_encoding = PositionalEncoding(args.embed_dim)   # (reference quirk: default dropout 0.1, not args.dropout)
        self.wdc = nn.Linear(args.embed_dim, args.vocab_size)
        self.dropout_layer = nn.Dropout(p=args.dropout)
        self.vocab_embedding.weight.data.uniform_(-0.1, 0.1)   # init_weights (:566-572)
        self.wdc.bias.data.fill_(0)
        self.wdc.weight.data.uniform_(-0.1, 0.1)

    def forward(self, memory, encoded_captions, caption_lengths):
        tgt = encoded_captions.permute(1, 0)
        x = self.position_encoding(self.vocab_embedding(tgt))
        # (the reference's mask is float32 whatever the module's dtype; in float32 the cast is the identity.  In float64 --
        # the yardstick evaluations of the tests -- a float32 mask makes nn.MultiheadAttention's fused path return garbage
        # without an error (need_weights=True raises "Input dtypes must be the same"): scores moved by 4.3, loss by 1 %)
        mask = causal_mask(tgt.size(0), tgt.device).to(x.dtype)
        for layer in self.transformer.layers:
            x = layer(x, memory, tgt_mask=mask)
        pred = self.wdc(self.dropout_layer(x)).permute(1, 0, 2)
        caption_lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True)
        encoded_captions = encoded_captions[sort_ind]
        pred = pred[sort_ind]
        decode_lengths = (caption_lengths - 1).tolist()
        return pred, encoded_captions, decode_lengths, sort_ind


def cc_loss(scores, caps_sorted, decode_lengths):
    """reference scripts/train_CC.py:124-132: packed CrossEntropyLoss(ignore_index=0) over the decoded steps."""
    targets = caps_sorted[:, 1:]
    s = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
    t = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
    return nn.functional.cross_entropy(s, t, ignore_index=0), s, t


def cc_forward_loss(trainer, imgs_a, imgs_b, caps, caplens):
    """One forward of the CC train step (reference scripts/train_CC.py:111-132): returns (loss, scores, targets,
    encoder feature)."""
    feat = trainer.update_cc(imgs_a, imgs_b)                        # (B, 192, 16, 16)
    B, C, H, W = feat.shape
    memory = feat.permute(2, 3, 0, 1).reshape(H * W, B, C)          # rearrange 'b c h w -> (h w) b c'
    scores, caps_sorted, decode_lengths, _ = trainer.decoder(memory, caps, caplens)
    loss, s, t = cc_loss(scores, caps_sorted, decode_lengths)
    return loss, s, t, feat


def clip_gradient(params, grad_clip):
    """reference model/utils.py:481-491."""
    for p in params:
        if p.grad is not None:
            p.grad.data.clamp_(-grad_clip, grad_clip)


def make_cc_optimizers(trainer, encoder_lr=1e-4, decoder_lr=1e-4):
    """reference scripts/train_CC.py:436-458 (both Adam, weight_decay 1e-5; StepLR(900, gamma=1) never changes lr)."""
    enc = torch.optim.Adam([p for p in trainer.encoder.parameters() if p.requires_grad], lr=encoder_lr, weight_decay=1e-5)
    dec = torch.optim.Adam([p for p in trainer.decoder.parameters() if p.requires_grad], lr=decoder_lr, weight_decay=1e-5)
    return enc, dec


def decoder_step_scores(decoder, k_prev_words, memory_bf):
    """One decoding pass of the reference's evaluation loop (scripts/train_CC.py:258-267) through `decoder`'s own
    sub-modules: k_prev_words int64 [s, 52], memory_bf [s, S, D] (beam first) -> scores [s, 52, vocab].  `decoder` may
    be this file's `CaptionDecoder` or the REAL reference module (the layers are looped for the torch-2.10 reason in
    the header); that is how `beam_search` below is pinned: identical captions through both."""
    tgt = k_prev_words.permute(1, 0)
    mask = causal_mask(tgt.size(0), tgt.device)
    x = decoder.position_encoding(decoder.vocab_embedding(tgt))
    enc = memory_bf.permute(1, 0, 2)
    for layer in decoder.transformer.layers:
        x = layer(x, enc, tgt_mask=mask)
    return decoder.wdc(x).permute(1, 0, 2)


def beam_search(decoder, encoder_out, start_id, end_id, beam_size, vocab_size, max_len=52, step_scores=None):
    """Restatement of the beam search inside `evaluate()` (reference scripts/train_CC.py:214-330) for ONE image pair.
    encoder_out: (S, 1, D) = rearrange(encoder(..., output_final=True), 'b c h w -> (h w) b c').
    Returns (best_seq or None, complete_seqs, complete_seqs_scores).

    Kept exactly: the whole 52-token window is re-decoded at every step (no key/value cache) and the scores are read
    at position step-1; at step 1 only beam 0 is expanded; hypotheses leave the beam when they emit <end>; the loop
    stops when the beam is empty or after step 51; the winner is the FIRST maximum of the completed scores; a pair
    whose beams never emit <end> yields no caption (`complete_inds` of the last step is empty, :326-328).
    `evaluate()` itself cannot be imported here (torchvision, cv2, h5py, skimage and the java-backed METEOR scorer
    are absent), so this loop's bookkeeping is pinned only through the decoder arithmetic (see decoder_step_scores):
    PARITY UNPINNED for the bookkeeping lines, stated in DESIGN.md."""
    import torch.nn.functional as F
    step_scores = step_scores or decoder_step_scores
    k = beam_size
    dev = encoder_out.device
    k_prev_words = torch.zeros(k, max_len, dtype=torch.int64, device=dev)
    k_prev_words[:, 0] = start_id
    seqs = torch.full((k, 1), start_id, dtype=torch.int64, device=dev)
    top_k_scores = torch.zeros(k, 1, device=dev)
    complete_seqs, complete_seqs_scores = [], []
    S, D = encoder_out.size(0), encoder_out.size(-1)
    enc = encoder_out.expand(S, k, D).permute(1, 0, 2)            # [k, S, D]
    step = 1
    with torch.no_grad():
        while True:
            scores = step_scores(decoder, k_prev_words, enc)       # [s, 52, V]
            scores = F.log_softmax(scores[:, step - 1, :], dim=1)
            scores = top_k_scores.expand_as(scores) + scores
            if step == 1:
                top_k_scores, top_k_words = scores[0].topk(k, 0, True, True)
            else:
                top_k_scores, top_k_words = scores.view(-1).topk(k, 0, True, True)
            prev_word_inds = top_k_words // vocab_size
            next_word_inds = top_k_words % vocab_size
            seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], dim=1)
            incomplete_inds = [i for i, w in enumerate(next_word_inds.tolist()) if w != end_id]
            complete_inds = sorted(set(range(len(next_word_inds))) - set(incomplete_inds))
            if complete_inds:
                complete_seqs.extend(seqs[complete_inds].tolist())
                complete_seqs_scores.extend(top_k_scores[complete_inds].tolist())
            k -= len(complete_inds)
            if k == 0:
                break
            seqs = seqs[incomplete_inds]
            enc = enc[prev_word_inds[incomplete_inds]]
            top_k_scores = top_k_scores[incomplete_inds].unsqueeze(1)
            k_prev_words = k_prev_words[incomplete_inds]
            k_prev_words[:, :step + 1] = seqs
            if step > 50:
                break
            step += 1
    if not complete_seqs_scores:
        return None, complete_seqs, complete_seqs_scores
    best = complete_seqs_scores.index(max(complete_seqs_scores))
    return complete_seqs[best], complete_seqs, complete_seqs_scores


def strip_special(seq, start_id, end_id, pad_id=0):
    """Hypothesis words as `evaluate()` stores them (reference scripts/train_CC.py:345)."""
    return [w for w in seq if w not in {start_id, end_id, pad_id}]


# (weight seed, beam size, embedding scale, <end> id) of tests/golden/cc_beam.npz.  Random decoder weights give nearly
# context-free logits; scaling the token embedding makes the hypotheses depend on their history, and <end> is a
# token those weights emit mid-sequence (found by decoding once without an <end>), so beams complete at different
# steps, shrink, and keep going.  The last case never emits its <end>: no caption (reference :326-328).
BEAM_CASES = ((6, 4, 10.0, 63), (4, 5, 30.0, 13), (6, 4, 10.0, 84), (6, 1, 10.0, 84), (3, 3, 30.0, 9), (4, 3, 30.0, 1))


def beam_case(seed, embed_scale=30.0, vocab=97, size=32):
    """Set-up shared by the beam-search tests and the fixture generator: CC Trainer (this oracle) with seeded synthetic
    weights; <start> = vocab-2, <pad> = 0.  Returns (args, trainer, state_dict, memory (S, 1, D))."""
    from . import model as om, synth
    args = synth.make_cc_args(size=size, vocab_size=vocab, dropout=0.0)
    ora = om.Trainer(args)
    sd = synth.synth_state_dict(ora, seed=seed)
    sd["decoder.position_encoding.pe"] = ora.state_dict()["decoder.position_encoding.pe"].clone()
    sd["decoder.vocab_embedding.weight"] = sd["decoder.vocab_embedding.weight"] * embed_scale
    ora.load_state_dict(sd)
    ora.eval()
    pre, post, _ = synth.synth_batch(1, size, seed=seed + 1)
    with torch.no_grad():
        feat = ora.update_cc(pre, post)
    B, C, H, W = feat.shape
    return args, ora, sd, feat.permute(2, 3, 0, 1).reshape(H * W, B, C)
