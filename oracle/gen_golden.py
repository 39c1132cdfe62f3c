"""ORACLE — golden-vector generator (build container only; needs /root/reference).

Runs the REAL reference (`model/trainer.py::Trainer.update_bcd`, `model/utils.py::
BCEDiceLoss`/`adjust_learning_rate`, `utils/metric_tool.py::get_confuse_matrix`/`cm2score`,
Adam as in `scripts/train_BCD.py:284-290`) on PyTorch-CPU fp32 with seeded synthetic
weights/inputs from `oracle/synth.py`, first asserting that the repo's restatement
(`oracle/model.py`) reproduces it bit-for-bit, then writes small fixtures:

    tests/golden/bcd_s{S}_b{B}.npz      (BCD: update_bcd, BCE+Dice, 3 Adam steps, eval mode)
    tests/golden/scd_s{64,256}_b{B}.npz (SCD, SURVEY.md 8(f).1: update_scd + the train_SCD.py loss)
    tests/golden/cc_s{S}_b{B}.npz       (CC, SURVEY.md 8(f).2: encoder blocks 0-4 + CaptionDecoder + packed CE,
                                         two Adam steps with gradient clipping as scripts/train_CC.py:118-147)

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden [--sizes 64 256]
The fixtures are DATA (inputs are regenerated from seeds; expected outputs are stored).
"""
import argparse
import os
import sys

import numpy as np
import torch

from . import model as om
from . import ref_import, synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
WEIGHT_SEED = 16
DATA_SEED = 0
MASK_MARGIN = 0.25
N_STEPS = 3
BASE_LR = 2e-4
MAX_ITER = 80000


def probe_idx(numel, n=64, seed=7):
    return np.random.default_rng(seed).integers(0, numel, size=n)


def summarize(t):
    t = t.detach().double().contiguous().view(-1)
    idx = probe_idx(t.numel())
    return np.concatenate([[t.mean().item(), t.std().item(), t.norm().item()], t[idx].numpy()])


def run(size, batch, check_restatement=True):
    tr, mu, met = ref_import.import_reference()
    args = om.make_args(size=size)
    args.lr_mode, args.lr, args.max_epochs, args.step_loss = "poly", BASE_LR, 1, 100
    ref = tr.Trainer(args)
    sd = synth.synth_state_dict(ref, seed=WEIGHT_SEED, mask_margin=MASK_MARGIN)
    ref.load_state_dict(sd, strict=True)
    pre, post, tgt = synth.synth_batch(batch, size, seed=DATA_SEED)
    out = {"meta": np.array([size, batch, WEIGHT_SEED, DATA_SEED, N_STEPS], dtype=np.int64),
           "mask_margin": np.array(MASK_MARGIN), "base_lr": np.array(BASE_LR),
           "max_iter": np.array(MAX_ITER)}

    # ---- eval-mode forward (BN running stats), reference scripts/train_BCD.py:92-154.
    # The synthetic running statistics are first replaced by the batch statistics of ONE
    # train-mode pass with momentum 1.0 (otherwise eval activations blow up and every
    # probability saturates, which would make the eval comparison vacuous).
    def calibrate(net, a, b):
        bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm3d)]
        net.train()
        for m in bns:
            m.momentum = 1.0
        with torch.no_grad():
            net.update_bcd(a, b)
        for m in bns:
            m.momentum = 0.1
        net.eval()

    calibrate(ref, pre, post)
    with torch.no_grad():
        p_eval = ref.update_bcd(pre, post)
        loss_eval = mu.BCEDiceLoss(p_eval, tgt)
    ref.load_state_dict(sd, strict=True)
    stride = max(1, size // 32)
    out["eval_prob_lattice"] = p_eval[:, :, ::stride, ::stride].numpy()
    out["eval_prob_full"] = p_eval.numpy().astype(np.float32) if size <= 64 else np.zeros(0, np.float32)
    out["eval_mask_bits"] = np.packbits((p_eval > 0.5).numpy().astype(np.uint8).reshape(-1))
    out["eval_band"] = np.array(int(((p_eval - 0.5).abs() < 1e-4).sum()))
    out["eval_loss"] = np.array(loss_eval.item())

    # ---- train-mode forward + backward (fresh module so BN buffers are untouched)
    ref.train()
    feats = ref.encoder(pre, post)
    for i, f in enumerate(feats):
        out[f"train_feat_c{i + 1}"] = summarize(f[0])
    opt = torch.optim.Adam(ref.parameters(), BASE_LR, (0.9, 0.99), eps=1e-08, weight_decay=1e-4)
    ref.load_state_dict(sd, strict=True)  # undo the BN running-stat update of the probe forward
    losses, lrs = [], []
    cm_total = np.zeros((2, 2))
    for it in range(N_STEPS):
        lr = mu.adjust_learning_rate(args, opt, 0, it, MAX_ITER, lr_factor=1.0)
        prob = ref.update_bcd(pre, post)
        loss = mu.BCEDiceLoss(prob, tgt)
        pred = torch.where(prob > 0.5, torch.ones_like(prob), torch.zeros_like(prob)).long()
        opt.zero_grad()
        loss.backward()
        if it == 0:
            out["train_prob_lattice"] = prob.detach()[:, :, ::stride, ::stride].numpy()
            out["train_prob_full"] = (prob.detach().numpy().astype(np.float32) if size <= 64
                                      else np.zeros(0, np.float32))
            out["train_mask_bits"] = np.packbits(pred.numpy().astype(np.uint8).reshape(-1))
            out["train_band"] = np.array(int(((prob - 0.5).abs() < 1e-4).sum()))
            names, norms, probes = [], [], []
            for n, p in ref.named_parameters():
                if p.grad is None:
                    continue
                names.append(n)
                norms.append(p.grad.double().norm().item())
                g = p.grad.detach().double().view(-1)
                probes.append(g[probe_idx(g.numel(), 4, seed=11)].numpy())
            ref_grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
            out["grad_names"] = np.array(names)
            out["grad_norms"] = np.array(norms)
            out["grad_probes"] = np.stack(probes)
            unused = [n for n, p in ref.named_parameters() if p.grad is None]
            out["unused_param_count"] = np.array(sum(dict(ref.named_parameters())[n].numel() for n in unused))
        opt.step()
        losses.append(loss.item())
        lrs.append(lr)
        cm_total += met.get_confuse_matrix(2, tgt.numpy(), pred.numpy())
    out["loss_curve"] = np.array(losses)
    out["lr_curve"] = np.array(lrs)
    out["cm_total"] = cm_total
    sc = met.cm2score(cm_total)
    out["scores"] = np.array([sc["Kappa"], sc["IoU"], sc["F1"], sc["OA"], sc["recall"], sc["precision"]])
    fin = ref.state_dict()
    out["final_param_l2"] = np.array([fin[n].double().norm().item() for n in out["grad_names"]])
    rm = [v.double().sum().item() for k, v in fin.items() if k.endswith("running_mean") and ".blocks.4." not in k and ".blocks.5." not in k]
    rv = [v.double().sum().item() for k, v in fin.items() if k.endswith("running_var") and ".blocks.4." not in k and ".blocks.5." not in k]
    out["final_running_mean_sums"] = np.array(rm)
    out["final_running_var_sums"] = np.array(rv)
    out["final_nbt"] = np.array([int(v) for k, v in fin.items() if k.endswith("num_batches_tracked")])

    # ---- float64 evaluation of the same reference modules: the yardstick for tolerances
    # (tests require |hip - f64| <= k * |reference_f32 - f64|, see tests/test_model_gpu.py)
    ref64 = tr.Trainer(args)
    ref64.load_state_dict(sd, strict=True)
    ref64 = ref64.double().train()
    opt64 = torch.optim.Adam(ref64.parameters(), BASE_LR, (0.9, 0.99), eps=1e-08, weight_decay=1e-4)
    pre64, post64, tgt64 = pre.double(), post.double(), tgt.double()
    losses64 = []
    for it in range(N_STEPS):
        mu.adjust_learning_rate(args, opt64, 0, it, MAX_ITER, lr_factor=1.0)
        prob64 = ref64.update_bcd(pre64, post64)
        loss64 = mu.BCEDiceLoss(prob64, tgt64)
        opt64.zero_grad()
        loss64.backward()
        if it == 0:
            out["train_prob_lattice_f64"] = prob64.detach()[:, :, ::stride, ::stride].numpy()
            out["train_prob_full_f64"] = prob64.detach().numpy() if size <= 64 else np.zeros(0)
            named64 = dict(ref64.named_parameters())
            out["grad_norms_f64"] = np.array([named64[str(n)].grad.norm().item() for n in out["grad_names"]])
            out["grad_probes_f64"] = np.stack([named64[str(n)].grad.detach().view(-1)[probe_idx(
                named64[str(n)].numel(), 4, seed=11)].numpy() for n in out["grad_names"]])
        opt64.step()
        losses64.append(loss64.item())
    out["loss_curve_f64"] = np.array(losses64)
    ref64.load_state_dict(sd, strict=True)
    ref64 = ref64.double()
    calibrate(ref64, pre64, post64)
    with torch.no_grad():
        pe64 = ref64.update_bcd(pre64, post64)
    out["eval_prob_lattice_f64"] = pe64[:, :, ::stride, ::stride].numpy()

    if check_restatement:
        ora = om.Trainer(om.make_args(size=size))
        ora.load_state_dict(sd, strict=True)
        ora.train()
        o = ora.update_bcd(pre, post)
        lo = om.bce_dice_loss(o, tgt)
        assert abs(lo.item() - losses[0]) == 0.0, (lo.item(), losses[0])
        assert np.array_equal(np.packbits(om.binarize(o).numpy().astype(np.uint8).reshape(-1)),
                              out["train_mask_bits"])
        assert abs(om.poly_lr(BASE_LR, 1, MAX_ITER, 0) - lrs[1]) < 1e-18
        lo.backward()
        ora_grads = {n: p.grad for n, p in ora.named_parameters() if p.grad is not None}
        assert set(ora_grads) == set(ref_grads)
        for n, g in ref_grads.items():
            assert torch.equal(g, ora_grads[n]), n
        print(f"[gen_golden] restatement == reference at size {size} (loss {losses[0]:.6f}, mask, all {len(ref_grads)} gradients bit-identical)")

    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, f"bcd_s{size}_b{batch}.npz")
    np.savez_compressed(path, **out)
    print(f"[gen_golden] wrote {path} ({os.path.getsize(path) / 1024:.1f} kB); losses {losses}")


def run_scd(size, batch):
    """SURVEY.md 8(f).1 fixture: the REAL reference `Trainer.update_scd` (K=3, T=5, three decoders, 7 classes)
    and the loss of `scripts/train_SCD.py:226-229` built from the reference's own `CrossEntropyLoss2d`,
    `ChangeSimilarity` and `BCEDiceLoss`; the restatement must reproduce both bit-for-bit."""
    tr, mu, _ = ref_import.import_reference()
    args = om.make_args(num_perception_frame=3, size=size, dataset="SECOND", num_class=7)
    ref = tr.Trainer(args)
    sd = synth.synth_state_dict(ref, seed=WEIGHT_SEED, mask_margin=MASK_MARGIN)
    ref.load_state_dict(sd, strict=True)
    pre, post, _ = synth.synth_batch(batch, size, seed=DATA_SEED)
    labels = synth.synth_scd_labels(batch, size, seed=DATA_SEED)
    seg_loss, sim_loss = mu.CrossEntropyLoss2d(ignore_index=0), mu.ChangeSimilarity()

    def ref_loss(net, a, b, dt):
        pm, qm, cm = net.update_scd(a, b)
        lc = labels[:, 2].long()
        pl, ql = labels[:, 0].long() * lc, labels[:, 1].long() * lc
        segm = seg_loss(pm, pl) + seg_loss(qm, ql)
        binary = mu.BCEDiceLoss(cm, lc.unsqueeze(1).to(dt))
        sim = sim_loss(pm[:, 1:], qm[:, 1:], lc.unsqueeze(1))
        return (pm, qm, cm), segm * 0.5 + binary + sim

    ref.train()
    outs, loss = ref_loss(ref, pre, post, torch.float32)
    loss.backward()
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    named = dict(ref.named_parameters())
    stride = max(size // 32, 1)
    out = {"meta": np.array([size, batch, WEIGHT_SEED, DATA_SEED, 3, 7], dtype=np.int64),
           "loss": np.array(loss.item()), "grad_names": np.array(names),
           "grad_norms": np.array([named[n].grad.norm().item() for n in names])}
    for k, o in zip(("pre", "post", "change"), outs):
        out[f"{k}_lattice"] = o.detach()[:, :, ::stride, ::stride].numpy()
    out["pre_argmax_bits"] = np.packbits((outs[0].detach().argmax(1) == labels[:, 0]).numpy().reshape(-1))
    ref64 = tr.Trainer(args)
    ref64.load_state_dict(sd, strict=True)
    ref64 = ref64.double().train()
    outs64, loss64 = ref_loss(ref64, pre.double(), post.double(), torch.float64)
    loss64.backward()
    named64 = dict(ref64.named_parameters())
    out["loss_f64"] = np.array(loss64.item())
    out["grad_norms_f64"] = np.array([named64[n].grad.norm().item() for n in names])
    for k, o in zip(("pre", "post", "change"), outs64):
        out[f"{k}_lattice_f64"] = o.detach()[:, :, ::stride, ::stride].numpy()
    # restatement == reference, bit for bit (outputs, loss, every gradient)
    ora = om.Trainer(om.make_args(num_perception_frame=3, size=size, dataset="SECOND", num_class=7))
    ora.load_state_dict(sd, strict=True)
    ora.train()
    oo = ora.update_scd(pre, post)
    lo = om.scd_loss(*oo, labels)
    lo.backward()
    assert all(torch.equal(a, b) for a, b in zip(oo, outs)) and lo.item() == loss.item(), (lo.item(), loss.item())
    on = dict(ora.named_parameters())
    assert all(torch.equal(on[n].grad, named[n].grad) for n in names)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, f"scd_s{size}_b{batch}.npz")
    np.savez_compressed(path, **out)
    print(f"[gen_golden] SCD restatement == reference; wrote {path} ({os.path.getsize(path) / 1024:.1f} kB); "
          f"loss {loss.item():.6f} (fp64 {loss64.item():.6f})")


CC_STEPS, CC_LR, CC_CLIP = 2, 1e-4, 5.0


def ref_cc_forward(net, pre, post, caps, caplens):
    """The reference CC forward (scripts/train_CC.py:111-132) through the REAL reference modules.  Two lines of the
    reference cannot run on torch 2.10 / CPU: `nn.TransformerDecoder.forward` passes
    `tgt_is_causal=` to the custom layer (TypeError) and `mask.cuda()`; the layers are therefore applied in a loop
    (there is no final norm: model/caption_decoder.py:555) with the mask built by the reference's own formula."""
    feat = net.update_cc(pre, post)
    B, C, H, W = feat.shape
    memory = feat.permute(2, 3, 0, 1).reshape(H * W, B, C)          # einops 'b c h w -> (h w) b c'
    dec = net.decoder
    tgt = caps.permute(1, 0)
    n = tgt.size(0)
    mask = (torch.triu(torch.ones(n, n)) == 1).transpose(0, 1)
    mask = mask.float().masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, float(0.0))
    x = dec.position_encoding(dec.vocab_embedding(tgt))
    # (identity in float32.  The float64 YARDSTICK evaluation needs it: with the reference's float32 mask on float64 queries
    # nn.MultiheadAttention's fused path silently returns garbage -- the round-1..4 fixtures' loss_f64 / grad_norms_f64 were 1 %
    # / 80-96 % away from the float32 values for that reason, not because the network is chaotic; regenerated in round 5)
    mask = mask.to(x.dtype)
    for layer in dec.transformer.layers:
        x = layer(x, memory, tgt_mask=mask)
    pred = dec.wdc(dec.dropout_layer(x)).permute(1, 0, 2)
    lens, sort_ind = caplens.squeeze(1).sort(dim=0, descending=True)
    caps_sorted, pred = caps[sort_ind], pred[sort_ind]
    decode_lengths = (lens - 1).tolist()
    from torch.nn.utils.rnn import pack_padded_sequence
    scores = pack_padded_sequence(pred, decode_lengths, batch_first=True).data
    targets = pack_padded_sequence(caps_sorted[:, 1:], decode_lengths, batch_first=True).data
    crit = torch.nn.CrossEntropyLoss(ignore_index=0)
    return crit(scores, targets), scores, targets, feat


def run_cc(size, batch):
    """SURVEY.md 8(f).2 fixture: reference encoder (`update_cc`, blocks 0-4) + reference `CaptionDecoder` modules,
    packed cross-entropy, gradient clipping and the two Adam optimisers of scripts/train_CC.py:436-458 (dropout 0 so
    that the result is deterministic); the restatement (oracle/caption.py) must reproduce all of it bit-for-bit."""
    import contextlib
    import io
    from . import caption as oc
    tr, mu, _ = ref_import.import_reference()
    args = synth.make_cc_args(size=size, dropout=0.0)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = tr.Trainer(args)
    ora = om.Trainer(args)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
    sd = synth.synth_state_dict(ref, seed=WEIGHT_SEED)
    sd["decoder.position_encoding.pe"] = ref.state_dict()["decoder.position_encoding.pe"].clone()   # constant table
    ref.load_state_dict(sd, strict=True)
    ora.load_state_dict(sd, strict=True)
    pre, post, _ = synth.synth_batch(batch, size, seed=DATA_SEED)
    caps, caplens = synth.synth_captions(batch, seed=DATA_SEED, vocab_size=args.vocab_size)
    ref.train(); ora.train()
    # reference quirk (model/caption_decoder.py:557): `PositionalEncoding(args.embed_dim)` keeps its DEFAULT dropout
    # 0.1 whatever --dropout says; it is switched off on the instances so that the fixture is deterministic
    ref.decoder.position_encoding.dropout.p = 0.0
    ora.decoder.position_encoding.dropout.p = 0.0
    enc_r, dec_r = oc.make_cc_optimizers(ref, CC_LR, CC_LR)
    enc_o, dec_o = oc.make_cc_optimizers(ora, CC_LR, CC_LR)
    out = {"meta": np.array([size, batch, WEIGHT_SEED, DATA_SEED, CC_STEPS, args.vocab_size], dtype=np.int64),
           "lr": np.array(CC_LR), "grad_clip": np.array(CC_CLIP)}
    losses = []
    for it in range(CC_STEPS):
        loss, scores, targets, feat = ref_cc_forward(ref, pre, post, caps, caplens)
        lo, so, to, fo = oc.cc_forward_loss(ora, pre, post, caps, caplens)
        assert torch.equal(feat, fo) and torch.equal(scores, so) and torch.equal(targets, to) and loss.item() == lo.item()
        for o in (dec_r, enc_r, dec_o, enc_o):
            o.zero_grad()
        loss.backward(); lo.backward()
        named, on = dict(ref.named_parameters()), dict(ora.named_parameters())
        names = [n for n, p in ref.named_parameters() if p.grad is not None]
        assert names == [n for n, p in ora.named_parameters() if p.grad is not None]
        assert all(torch.equal(on[n].grad, named[n].grad) for n in names)
        if it == 0:
            stride = max(feat.shape[-1] // 8, 1)
            out["feat_lattice"] = feat.detach()[:, :, ::stride, ::stride].numpy()
            out["feat_summary"] = summarize(feat)
            out["scores_summary"] = summarize(scores)
            out["scores_rows"] = scores.detach()[::max(scores.shape[0] // 16, 1)].numpy()
            out["targets"] = targets.numpy()
            out["grad_names"] = np.array(names)
            out["grad_norms"] = np.array([named[n].grad.norm().item() for n in names])
            out["unused_param_count"] = np.array(sum(p.numel() for p in ref.parameters() if p.grad is None))
            top1 = (scores.argmax(1) == targets).float().mean().item()
            out["top1"] = np.array(top1)
        for net, (eo, do) in ((ref, (enc_r, dec_r)), (ora, (enc_o, dec_o))):
            mu.clip_gradient(do, CC_CLIP)
            mu.clip_gradient(eo, CC_CLIP)
            eo.step(); do.step()
        losses.append(loss.item())
    out["loss_curve"] = np.array(losses)
    fin, fo = ref.state_dict(), ora.state_dict()
    assert all(torch.equal(fin[k], fo[k]) for k in fin)
    out["final_param_l2"] = np.array([fin[str(n)].double().norm().item() for n in out["grad_names"]])
    # fp64 yardstick (same reference modules)
    with contextlib.redirect_stdout(io.StringIO()):
        ref64 = tr.Trainer(args)
    ref64.load_state_dict(sd, strict=True)
    ref64 = ref64.double().train()
    ref64.decoder.position_encoding.dropout.p = 0.0
    l64, s64, _, f64 = ref_cc_forward(ref64, pre.double(), post.double(), caps, caplens)
    l64.backward()
    n64 = dict(ref64.named_parameters())
    stride = max(f64.shape[-1] // 8, 1)
    out["loss_f64"] = np.array(l64.item())
    out["feat_lattice_f64"] = f64.detach()[:, :, ::stride, ::stride].numpy()
    out["scores_rows_f64"] = s64.detach()[::max(s64.shape[0] // 16, 1)].numpy()
    out["grad_norms_f64"] = np.array([n64[str(n)].grad.norm().item() for n in out["grad_names"]])
    # (round 5) the float64 loss CURVE: how far the float32 reference's later losses drift from float64 once clipped-Adam
    # steps (whose first update is lr * sign(g): a sign decided by rounding on near-zero gradients) have been applied --
    # the yardstick for an implementation's step-2 loss, which the first loss alone does not give
    enc64, dec64 = oc.make_cc_optimizers(ref64, CC_LR, CC_LR)
    losses64 = [l64.item()]
    for it in range(1, CC_STEPS + 1):
        mu.clip_gradient(dec64, CC_CLIP)
        mu.clip_gradient(enc64, CC_CLIP)
        enc64.step(); dec64.step()
        if it == CC_STEPS:
            break
        dec64.zero_grad(); enc64.zero_grad()
        l64b = ref_cc_forward(ref64, pre.double(), post.double(), caps, caplens)[0]
        l64b.backward()
        losses64.append(l64b.item())
    out["loss_curve_f64"] = np.array(losses64)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, f"cc_s{size}_b{batch}.npz")
    np.savez_compressed(path, **out)
    print(f"[gen_golden] CC restatement == reference (encoder feature, logits, loss, every gradient, {CC_STEPS} clipped "
          f"Adam steps); wrote {path} ({os.path.getsize(path) / 1024:.1f} kB); loss curve {losses} (fp64 first {l64.item():.6f})")


def run_cc_beam():
    """tests/golden/cc_beam.npz: the beam search of reference scripts/train_CC.py:214-330 (restated in
    oracle/caption.py::beam_search -- `evaluate()` itself is not importable here) with every decoding step running
    through the REAL reference `CaptionDecoder` sub-modules, on the seeded cases of oracle/caption.py::BEAM_CASES."""
    import contextlib
    import io
    import numpy as np
    from . import caption as oc
    tr, _, _ = ref_import.import_reference()
    best = np.zeros((len(oc.BEAM_CASES), 53), dtype=np.int64)
    best_len = np.zeros(len(oc.BEAM_CASES), dtype=np.int64)
    scores = np.full((len(oc.BEAM_CASES), 8), np.nan, dtype=np.float64)
    for i, (seed, beam, es, end_id) in enumerate(oc.BEAM_CASES):
        args, ora, sd, memory = oc.beam_case(seed, es)
        with contextlib.redirect_stdout(io.StringIO()):
            ref = tr.Trainer(args)
        ref.load_state_dict(sd)
        ref.eval()
        V = args.vocab_size
        b, seqs, sc = oc.beam_search(ref.decoder, memory, V - 2, end_id, beam, V)
        b2, seqs2, sc2 = oc.beam_search(ora.decoder, memory, V - 2, end_id, beam, V)
        assert b == b2 and seqs == seqs2 and sc == sc2
        best_len[i] = len(b or [])
        best[i, :best_len[i]] = b or []
        scores[i, :len(sc)] = sc
        print(f"[gen_golden] beam case seed={seed} k={beam}: complete lens {[len(s) for s in seqs]}, best len {best_len[i]}, "
              f"scores {[round(s, 3) for s in sc]}")
    path = os.path.join(GOLDEN_DIR, "cc_beam.npz")
    np.savez_compressed(path, cases=np.array(oc.BEAM_CASES, dtype=np.float64), best=best, best_len=best_len, scores=scores)
    print(f"[gen_golden] wrote {path}")


def run_metrics():
    """SCD / CC validation metrics through the REAL reference functions (model/utils.py: accuracy, SCDD_eval_all,
    caption_accuracy, AverageMeter) on seeded label maps / score matrices -> tests/golden/scd_metrics.npz; the
    restatement (oracle/metrics.py) is asserted identical first."""
    from oracle import metrics as om_
    _, mu, _ = ref_import.import_reference()
    rng = np.random.default_rng(7)
    nc, n_img, S = 7, 6, 48
    labels = [rng.integers(0, nc, size=(S, S)).astype(np.int64) * (rng.random((S, S)) < 0.4) for _ in range(n_img)]
    preds = [np.where(rng.random((S, S)) < 0.7, l, rng.integers(0, nc, size=(S, S))).astype(np.int64) for l in labels]
    ref_scores = mu.SCDD_eval_all(preds, labels, nc)
    ora_scores = om_.SCDD_eval_all(preds, labels, nc)
    assert tuple(float(v) for v in ref_scores) == tuple(float(v) for v in ora_scores), (ref_scores, ora_scores)
    accs = np.array([mu.accuracy(p, l)[0] for p, l in zip(preds, labels)])
    assert np.array_equal(accs, np.array([om_.accuracy(p, l)[0] for p, l in zip(preds, labels)]))
    accs_nz = np.array([mu.accuracy(p, l, ignore_zero=True)[0] for p, l in zip(preds, labels)])
    hist = np.zeros((nc, nc))
    for p, l in zip(preds, labels):
        hist += mu.get_hist(p, l, nc)
    m_ref, m_ora = mu.AverageMeter(), om_.AverageMeter()
    for v in accs:
        m_ref.update(float(v)); m_ora.update(float(v))
    assert m_ref.average() == m_ora.average()
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(97, 53, generator=g)
    targets = torch.randint(0, 53, (97,), generator=g)
    scores[torch.arange(0, 97, 3), targets[::3]] += 6.0
    cap = np.array([mu.caption_accuracy(scores, targets, k) for k in (1, 5)])
    assert np.array_equal(cap, np.array([om_.caption_accuracy(scores, targets, k) for k in (1, 5)]))
    path = os.path.join(GOLDEN_DIR, "scd_metrics.npz")
    np.savez_compressed(path, preds=np.stack(preds).astype(np.int8), labels=np.stack(labels).astype(np.int8), num_class=np.int64(nc),
                        scores=np.array(ref_scores, dtype=np.float64), hist=hist, acc=accs, acc_ignore_zero=accs_nz,
                        acc_meter=np.float64(m_ref.average()), cap_scores=scores.numpy(), cap_targets=targets.numpy(), cap_acc=cap)
    print(f"[gen_golden] wrote {path}: Fscd {ref_scores[0]:.6f} mIoU {ref_scores[1]:.6f} SeK {ref_scores[2]:.6f}, "
          f"caption top-1/5 {cap[0]:.3f}/{cap[1]:.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--scd-only", action="store_true", help="only regenerate the SCD fixture")
    ap.add_argument("--cc-only", action="store_true", help="only regenerate the CC fixtures")
    ap.add_argument("--cc-beam", action="store_true", help="only regenerate the CC beam-search fixture")
    ap.add_argument("--metrics", action="store_true", help="only regenerate the SCD / CC validation-metric fixture")
    a = ap.parse_args()
    sys.dont_write_bytecode = True
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    if a.cc_beam:
        return run_cc_beam()
    if a.metrics:
        return run_metrics()
    if not a.scd_only and not a.cc_only:
        for s in a.sizes:
            run(s, a.batch)
    if not a.cc_only:
        for s in (64, 256):      # the benchmarked SCD resolution has a reference-generated fixture too
            if s in a.sizes or a.scd_only:
                run_scd(s, a.batch)
    if not a.scd_only:
        for s in (64, 256):
            run_cc(s, a.batch)


if __name__ == "__main__":
    main()
