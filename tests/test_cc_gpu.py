"""Change-captioning path on the GPU (BASELINE.json configs[4]): X3D blocks 0-4 (res5 runs the wide
GEMM kernels) + the caption decoder (csrc/caption_ops.hip) + packed cross-entropy + clipped two-optimizer Adam step,
against the oracle restatement (oracle/caption.py) and the fixtures produced by the REAL reference modules
(tests/golden/cc_s{64,256}_b2.npz, oracle/gen_golden.py::run_cc).  Dropout is 0 in the parity cases (torch's CPU dropout
stream cannot be reproduced on the device); a separate test checks the train-mode dropout statistics."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K_NOISE = 4.0


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _mirror(args, sd):
    from change3d_amd.model.trainer import Trainer
    with contextlib.redirect_stdout(io.StringIO()):
        net = Trainer(args)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.decoder.position_encoding.dropout.p = 0.0   # reference quirk switched off, as in the fixture (gen_golden.run_cc)
    return net


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_caption_decoder_module_vs_oracle(dtype, tol):
    """Decoder alone (embedding .. logits, packed CE, every gradient incl. the memory gradient) on a random memory."""
    _need_gpu()
    from oracle import caption as oc
    from change3d_amd import synthetic as synth
    from change3d_amd.model.caption_decoder import CaptionDecoder, packed_cross_entropy
    args = synth.make_cc_args(size=64, vocab_size=203, dropout=0.0)
    args.act_dtype = dtype
    ref = oc.CaptionDecoder(args)
    sd = synth.synth_state_dict(ref, seed=9)
    sd["position_encoding.pe"] = ref.state_dict()["position_encoding.pe"].clone()
    ref.load_state_dict(sd)
    ref.train()
    ref.position_encoding.dropout.p = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        mine = CaptionDecoder(args)
    mine.load_state_dict(sd)
    mine = mine.to(DEV).train()
    mine.position_encoding.dropout.p = 0.0
    S, B = 37, 3
    mem = synth.synth_tensor((S, B, 192), 11)
    caps, caplens = synth.synth_captions(B, seed=4, vocab_size=203)
    mr = mem.clone().requires_grad_(True)
    scores, caps_sorted, dl, _ = ref(mr, caps, caplens)
    lr_, s_ref, _ = oc.cc_loss(scores, caps_sorted, dl)
    lr_.backward()
    md = mem.to(DEV).requires_grad_(True)
    lg = mine.logits_seq_first(md, caps.to(DEV))
    ld = packed_cross_entropy(lg, caps.to(DEV), caplens.to(DEV), 203)
    ld.backward()
    torch.cuda.synchronize()
    # API-level forward too (sorted (B, L, V) predictions)
    with torch.no_grad():
        pred, cs, dl2, si = mine(md, caps.to(DEV), caplens.to(DEV))
    assert dl2 == dl and torch.equal(cs.cpu(), caps_sorted)
    assert (pred.cpu() - scores.detach()).abs().max().item() < tol * max(1.0, scores.abs().max().item())
    assert abs(ld.item() - lr_.item()) < tol * max(1.0, abs(lr_.item()))
    assert rel(md.grad, mr.grad) < 10 * tol, ("memory gradient", rel(md.grad, mr.grad))
    pr = dict(ref.named_parameters())
    used = {id(p) for p in mine.used_parameters()}
    worst = []
    for n, p in mine.named_parameters():
        if id(p) in used:
            assert p.grad is not None and pr[n].grad is not None, n
            r = rel(p.grad, pr[n].grad)
            if r > 10 * tol:
                worst.append((n, r))
        else:
            assert p.grad is None and pr[n].grad is None, n
    assert not worst, worst


@pytest.mark.parametrize("size", [64, 256])
def test_e2e_cc_vs_reference_golden(size, golden_dir):
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.caption_decoder import packed_cross_entropy
    from change3d_amd.model.utils import FusedAdam, ParamArena, cc_named_params, clip_gradient
    from test_model_gpu import _noise_check
    G = np.load(os.path.join(golden_dir, f"cc_s{size}_b2.npz"))
    batch, vocab = int(G["meta"][1]), int(G["meta"][5])
    args = synth.make_cc_args(size=size, vocab_size=vocab, dropout=0.0)
    from oracle import model as om
    ora = om.Trainer(args)
    sd = synth.synth_state_dict(ora, seed=int(G["meta"][2]))
    sd["decoder.position_encoding.pe"] = ora.state_dict()["decoder.position_encoding.pe"].clone()
    net = _mirror(args, sd)
    pre, post, _ = (t.to(DEV) for t in synth.synth_batch(batch, size, seed=int(G["meta"][3])))
    caps, caplens = (t.to(DEV) for t in synth.synth_captions(batch, seed=int(G["meta"][3]), vocab_size=vocab))
    enc_named, dec_named = cc_named_params(net)
    enc_arena, dec_arena = ParamArena(enc_named, torch.device(DEV)), ParamArena(dec_named, torch.device(DEV))
    lr = float(G["lr"])
    enc_opt = FusedAdam(enc_arena, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    dec_opt = FusedAdam(dec_arena, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    losses = []
    for it in range(int(G["meta"][4])):
        feat = net.update_cc(pre, post)
        B, C, H, W = feat.shape
        memory = feat.permute(2, 3, 0, 1).reshape(H * W, B, C)
        lg = net.decoder.logits_seq_first(memory, caps)
        loss = packed_cross_entropy(lg, caps, caplens, vocab)
        dec_opt.zero_grad(); enc_opt.zero_grad()
        loss.backward()
        if it == 0:
            stride = max(feat.shape[-1] // 8, 1)
            lat = feat.detach()[:, :, ::stride, ::stride].cpu().double().numpy()
            e_ref = np.abs(G["feat_lattice"].astype(np.float64) - G["feat_lattice_f64"]).max()
            e_hip = np.abs(lat - G["feat_lattice_f64"]).max()
            print(f"CC s{size}: encoder feature max|x - x_fp64| hip {e_hip:.3e}  fp32 reference {e_ref:.3e}")
            assert e_hip <= K_NOISE * e_ref + 1e-5
            l_ref, l64 = float(G["loss_curve"][0]), float(G["loss_f64"])
            print(f"CC s{size}: loss hip {loss.item():.6f} ref32 {l_ref:.6f} ref64 {l64:.6f}")
            assert abs(loss.item() - l64) <= K_NOISE * abs(l_ref - l64) + 1e-3
            named = dict(net.named_parameters())
            names = [str(n) for n in G["grad_names"]]
            assert set(names) == {n for n, _ in enc_named + dec_named}
            gn = np.array([named[n].grad.double().norm().item() for n in names])
            eh = np.abs(gn - G["grad_norms_f64"]) / (G["grad_norms_f64"] + 1e-30)
            er = np.abs(G["grad_norms"] - G["grad_norms_f64"]) / (G["grad_norms_f64"] + 1e-30)
            _noise_check(G["grad_names"], eh, er, f"CC s{size} grad-norm rel err")
            unused = sum(p.numel() for p in net.parameters() if p.grad is None)
            assert unused == int(G["unused_param_count"])
        clip_gradient(dec_opt, float(G["grad_clip"]))
        clip_gradient(enc_opt, float(G["grad_clip"]))
        enc_opt.step(); dec_opt.step()
        losses.append(loss.item())
    print(f"CC s{size}: loss curve hip {losses} ref32 {G['loss_curve']}")
    # every step's loss against the float64 reference curve, with the float32 reference's own distance from it as the yardstick
    # (after one clipped-Adam step -- lr * sign(g) on near-zero gradients -- the float32 reference itself is 1.3e-2 (64 x 64) /
    # 2.0e-1 (256 x 256) away from float64 at step 2; the first loss alone, 9e-5, says nothing about that)
    for i, l in enumerate(losses):
        l32, l64 = float(G["loss_curve"][i]), float(G["loss_curve_f64"][i])
        assert abs(l - l64) <= K_NOISE * abs(l32 - l64) + 5e-3, (i, l, l32, l64)


def test_caption_decoder_dropout_is_reproducible_and_unbiased():
    """Train-mode dropout (p = 0.1 everywhere + the fixed 0.1 of the position encoding): same torch seed -> identical
    logits, different seed -> different; the mean of the dropped embedding matches the undropped one."""
    _need_gpu()
    from change3d_amd import ops, synthetic as synth
    from change3d_amd.model.caption_decoder import CaptionDecoder
    args = synth.make_cc_args(size=64, vocab_size=101, dropout=0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        dec = CaptionDecoder(args).to(DEV).train()
    mem = synth.synth_tensor((16, 2, 192), 3).to(DEV)
    caps, _ = synth.synth_captions(2, seed=1, vocab_size=101)
    caps = caps.to(DEV)
    torch.manual_seed(5)
    a = dec.logits_seq_first(mem, caps).detach().clone()
    torch.manual_seed(5)
    b = dec.logits_seq_first(mem, caps).detach().clone()
    torch.manual_seed(6)
    c = dec.logits_seq_first(mem, caps).detach().clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    x = torch.ones((4096, 192), device=DEV)
    y = torch.empty_like(x)
    ops.cap_dropout(x, y, 4096, 192, 0.1, 1234, ops.DT_F32)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 5e-3 and abs(y.mean().item() - 1.0) < 5e-3
    assert torch.all((y == 0) | ((y - 1.0 / 0.9).abs() < 1e-6))


@pytest.mark.parametrize("seed", [23, 21, 25])
def test_e2e_cc_vs_oracle_conditioned_weights(seed):
    """With the default synthetic weights the 70-block CC encoder is ill-conditioned (the fp32 reference's own gradient norms
    sit 0.1 % (median) .. 10 % (worst) away from its fp64 evaluation: the fixture test above judges against that yardstick --
    NOTE: until round 5 the fixtures' fp64 yardstick itself was wrong, 80-96 %, because the reference's float32 attention mask
    makes nn.MultiheadAttention's fused path return garbage on float64 queries; see oracle/gen_golden.py::ref_cc_forward).
    With every residual branch scaled by 0.1 (`branch_gain`, a trained-network-like stack) rounding stays in the linear
    regime and EVERY gradient of the whole path -- encoder blocks 0-4 through the caption decoder -- is compared
    parameter by parameter with the fp32 oracle, to 1e-4 relative L2, on three weight seeds.

    Kinks.  At 64x64 the res4 / res5 BatchNorms see 384 / 96 values per channel and some ReLU pre-activation of the 70 blocks
    sits within f32 noise of zero for most weight draws: which side it falls on depends on the summation order of the
    statistics, and ONE flipped unit moves that block's gradients by ~1e-3 and everything below it by ~1e-4.  Until round 5
    only seed 23 (off the kink under the kernels' exact roundings) carried the strict bound; seeds 21 (res5 block 1 flips)
    and 25 (res3 block 5 flips) had a loose one.  Now all three are strict: `oracle/kinks.py::strict_compare` names the
    at-risk units the HIP path took on the other side (printed) and requires every tensor to meet 1e-4 against the oracle
    with exactly those units flipped."""
    _need_gpu()
    from oracle import caption as oc, kinks, model as om
    from change3d_amd import synthetic as synth
    from change3d_amd.model.caption_decoder import packed_cross_entropy
    from change3d_amd.model.utils import cc_named_params
    size, batch, vocab = 64, 2, 157
    args = synth.make_cc_args(size=size, vocab_size=vocab, dropout=0.0)
    ora0 = om.Trainer(args)
    sd = synth.synth_state_dict(ora0, seed=seed, branch_gain=0.1)
    sd["decoder.position_encoding.pe"] = ora0.state_dict()["decoder.position_encoding.pe"].clone()
    net = _mirror(args, sd)
    pre, post, _ = synth.synth_batch(batch, size, seed=3)
    caps, caplens = synth.synth_captions(batch, seed=3, vocab_size=vocab)
    feat = net.update_cc(pre.to(DEV), post.to(DEV))
    B, C, H, W = feat.shape
    lg = net.decoder.logits_seq_first(feat.permute(2, 3, 0, 1).reshape(H * W, B, C), caps.to(DEV))
    loss = packed_cross_entropy(lg, caps.to(DEV), caplens.to(DEV), vocab)
    loss.backward()
    torch.cuda.synchronize()
    enc_named, dec_named = cc_named_params(net)
    g_hip = {n: p.grad.detach().cpu() for n, p in enc_named + dec_named}

    def make_run(dtype):
        ora = om.Trainer(args)
        ora.load_state_dict(sd)
        ora = (ora.double() if dtype == torch.float64 else ora).train()
        ora.decoder.position_encoding.dropout.p = 0.0

        def run():
            ora.zero_grad(set_to_none=True)
            lo, so, _, fo = oc.cc_forward_loss(ora, pre.to(dtype), post.to(dtype), caps, caplens)
            lo.backward()
            return [fo, lo]
        return ora, run

    def check(outs):
        fo, lo = outs
        print(f"conditioned CC: feature rel-L2 {rel(feat, fo):.2e}, loss hip {loss.item():.6f} oracle {lo.item():.6f}")
        assert rel(feat, fo) < 1e-4 and abs(loss.item() - lo.item()) < 1e-4

    errs, granted = kinks.strict_compare(g_hip, make_run, tol=1e-4, check_outputs=check)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    med = float(np.median(list(errs.values())))
    print(f"conditioned CC seed {seed}: worst per-parameter gradient rel-L2 {[(n, f'{e:.1e}') for n, e in worst]}; median {med:.1e}; "
          f"ReLU units granted the other side: {[(u[0], u[1], f'{u[3]:.2f} sigma') for u in granted]}")
    if worst[0][1] >= 1e-4 and os.path.isdir("gpurun_out"):
        torch.save(g_hip, f"gpurun_out/kink_fail_cc_{seed}.pt")
    assert med < 2e-5 and worst[0][1] < 1e-4 and len(granted) <= 8, (med, worst, granted)
    assert all(u[3] < 6.0 for u in granted), ("a granted unit must lie within 6 sigma of zero", granted)


def _beam_decoder(sd, args, dtype=torch.float32):
    from change3d_amd.model.caption_decoder import CaptionDecoder
    args.act_dtype = dtype
    with contextlib.redirect_stdout(io.StringIO()):
        dec = CaptionDecoder(args)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    return dec.to(DEV).eval()


def test_beam_search_vs_oracle_and_reference_fixture(golden_dir):
    """`CaptionDecoder.beam_search` (truncated window, cached memory projections, HIP kernels) decodes exactly the
    captions of the restated reference loop (oracle/caption.py::beam_search, 52-token window at every step) and of
    tests/golden/cc_beam.npz, whose steps ran through the REAL reference modules; scores to fp32 noise."""
    _need_gpu()
    from oracle import caption as oc
    g = np.load(os.path.join(golden_dir, "cc_beam.npz"))
    for i, (seed, beam, es, end_id) in enumerate(oc.BEAM_CASES):
        args, ora, sd, memory = oc.beam_case(seed, es)
        V = args.vocab_size
        dec = _beam_decoder(sd, args)
        want = oc.beam_search(ora.decoder, memory, V - 2, end_id, beam, V)
        got = dec.beam_search(memory.to(DEV), V - 2, end_id, beam)
        assert got[0] == want[0], (seed, beam, got[0], want[0])
        assert got[1] == want[1]
        assert np.allclose(got[2], want[2], rtol=0, atol=2e-3), (got[2], want[2])
        n = int(g["best_len"][i])
        assert (got[0] or []) == g["best"][i, :n].tolist()
        assert dec.training is False


def test_reference_evaluate_loop_runs_on_the_mirror_modules():
    """The reference's evaluation loop addresses the decoder's sub-modules one by one (scripts/train_CC.py:260-266:
    `vocab_embedding`, `position_encoding`, `transformer(tgt, memory, tgt_mask=mask)`, `wdc`): driven that way (the
    oracle's loop with the reference's call sequence as the step function) the mirror decodes the oracle's captions."""
    _need_gpu()
    from oracle import caption as oc

    def reference_call_sequence(decoder, k_prev_words, enc_bf):
        tgt = k_prev_words.permute(1, 0)
        mask = oc.causal_mask(tgt.size(0)).to(DEV)
        emb = decoder.position_encoding(decoder.vocab_embedding(tgt))
        pred = decoder.transformer(emb, enc_bf.permute(1, 0, 2), tgt_mask=mask)
        return decoder.wdc(pred).permute(1, 0, 2)

    for seed, beam, es, end_id in oc.BEAM_CASES[1:4]:
        args, ora, sd, memory = oc.beam_case(seed, es)
        V = args.vocab_size
        dec = _beam_decoder(sd, args)
        want = oc.beam_search(ora.decoder, memory, V - 2, end_id, beam, V)
        got = oc.beam_search(dec, memory.to(DEV), V - 2, end_id, beam, V, step_scores=reference_call_sequence)
        assert got[0] == want[0] and got[1] == want[1]
        assert np.allclose(got[2], want[2], rtol=0, atol=2e-3)
    with pytest.raises(NotImplementedError):
        dec.transformer(torch.zeros(4, 1, 192, device=DEV), memory.to(DEV), tgt_mask=torch.zeros(4, 4, device=DEV))


def test_beam_search_bf16_runs_and_terminates():
    """bf16 activations: no parity claim on the argmax path (near-ties flip), but every caption is well formed."""
    _need_gpu()
    from oracle import caption as oc
    seed, beam, es, end_id = oc.BEAM_CASES[2]
    args, ora, sd, memory = oc.beam_case(seed, es)
    V = args.vocab_size
    dec = _beam_decoder(sd, args, torch.bfloat16)
    best, seqs, scores = dec.beam_search(memory.to(DEV), V - 2, end_id, beam)
    assert len(seqs) == len(scores) <= beam
    for s in seqs:
        assert s[0] == V - 2 and s[-1] == end_id and end_id not in s[1:-1] and len(s) <= 53
    assert best is None or best in seqs


def test_packed_cross_entropy_rejects_out_of_range_targets():
    """A counted target outside [0, vocab) (wrong --vocab_size, corrupt word map) must not be dropped quietly:
    torch.nn.CrossEntropyLoss (reference scripts/train_CC.py:131-132) trips a device assert; the fused kernel turns the
    loss NaN.  ignore_index and the steps beyond a caption's length are still ignored."""
    _need_gpu()
    from change3d_amd.model.caption_decoder import packed_cross_entropy
    B, L, V = 3, 7, 11
    g = torch.Generator().manual_seed(5)
    Vp = (V + 7) // 8 * 8
    logits = torch.randn(L, B, Vp, generator=g).to(DEV)
    caps = torch.randint(1, V, (B, L), generator=g)
    caplens = torch.tensor([[7], [5], [4]])
    ok = packed_cross_entropy(logits, caps.to(DEV), caplens.to(DEV), V)
    assert torch.isfinite(ok).item()
    bad = caps.clone(); bad[1, 2] = V          # a decoded step of caption 1
    assert torch.isnan(packed_cross_entropy(logits, bad.to(DEV), caplens.to(DEV), V)).item()
    bad = caps.clone(); bad[2, 3] = -4
    assert torch.isnan(packed_cross_entropy(logits, bad.to(DEV), caplens.to(DEV), V)).item()
    pad = caps.clone(); pad[2, 5] = V + 3      # beyond caption 2's length: not counted, not an error
    same = packed_cross_entropy(logits, pad.to(DEV), caplens.to(DEV), V)
    assert torch.equal(same, ok)
