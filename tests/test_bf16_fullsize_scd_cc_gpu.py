"""GPU tests that pin the BENCHMARKED SCD and CC configurations (BASELINE.json configs[3], configs[4]): bf16 activations,
B=16 per GPU, 256x256 pairs -- SCD with T=5 (three perception frames, three decoders, 7 classes), CC with X3D blocks 0-4
(res5 on the wide GEMM kernels) + the caption decoder.  Parity proper is established at 64^2 / 256^2 B=2 in f32
(tests/test_model_gpu.py, tests/test_cc_gpu.py); here the throughput path is checked AT THE SIZE bench.py quotes,
through size-independent properties, exactly as tests/test_bf16_fullsize_gpu.py does for BCD:

  * one full train step is finite and run-to-run reproducible (activations / logits bit-identical; gradients up to the
    f32 atomics of the leaf-gradient kernels);
  * every decoder tap (and, for CC, the res5 encoder feature) of the bf16 path tracks the f32 HIP path -- the
    parity-tested path -- on identical weights and inputs within stated bf16 storage bounds (conditioned weights:
    `branch_gain` 0.1, a trained-network-like stack whose rounding errors stay in the linear regime);
  * the flat gradient of the bf16 step points the same way as the f32 one (cosine);
  * T=5 / res5 BatchNorm statistics at B=16 against an f64 recomputation from the stored bf16 activations.

Reference: model/trainer.py:243-266 (update_scd), :292-306 (update_cc); scripts/train_SCD.py:216-233; scripts/train_CC.py:111-145."""
import contextlib
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, S = 16, 256

# bf16-vs-f32 HIP relative L2 of the per-stage taps with conditioned weights.  BCD at B=32 measured 3.1e-3 / 5.6e-3 /
# 1.1e-2 / 2.0e-2 behind 4 / 19 / 49 / 124 stored tensors (tests/test_bf16_fullsize_gpu.py, bounds there 1e-2 / 2e-2 /
# 3e-2 / 5e-2); T=5 stores the same number of tensors per tap, res5 adds 60 more behind the CC feature.
TAP_BOUND = (1.0e-2, 2.0e-2, 3.0e-2, 5.0e-2)
RES5_BOUND = 8.0e-2


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _grad_repro(arena, g0, g1):
    worst = 0.0
    for p, o in zip(arena.params, arena.offsets):
        a, b = g0[o:o + p.numel()].double(), g1[o:o + p.numel()].double()
        worst = max(worst, ((a - b).norm() / (a.norm() + 1e-30)).item())
    return worst


# ------------------------------------------------------------------------------------------------- SCD
def _build_scd(act_dtype):
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    args = synth.make_args(num_perception_frame=3, size=S, dataset="SECOND", num_class=7)
    args.act_dtype = act_dtype
    with contextlib.redirect_stdout(io.StringIO()):
        net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25, branch_gain=0.1))
    return net.to(DEV).train()


def _scd_step(net, arena, pre, post, labels, taps=None):
    from change3d_amd.model.utils import ChangeSimilarity, CrossEntropyLoss2d
    from change3d_amd.scripts.train_SCD import scd_loss
    arena.zero_grad()
    if taps is not None:
        feats = net.encoder(pre, post)
        taps.extend(torch.stack([t.detach().float() for t in f]).clone() for f in feats)   # (K, B, C, H, W) per stage
    masks = net.update_scd(pre, post)
    loss = scd_loss(CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity(), masks, labels)[0]
    loss.backward()
    torch.cuda.synchronize()
    return [m.detach() for m in masks], loss.detach()


def test_scd_bf16_b16_256_step_finite_reproducible_and_tracks_f32():
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.utils import ParamArena, hot_path_named_params
    pre, post, _ = (t.to(DEV) for t in synth.synth_batch(B, S, seed=0))
    labels = synth.synth_scd_labels(B, S, seed=0).to(DEV)
    net = _build_scd(torch.bfloat16)
    arena = ParamArena(hot_path_named_params(net), torch.device(DEV))
    st0 = {k: v.clone() for k, v in net.state_dict().items()}
    runs = []
    for r in range(2):
        net.load_state_dict(st0)
        taps = [] if r == 0 else None
        masks, loss = _scd_step(net, arena, pre, post, labels, taps)
        runs.append(([m.clone() for m in masks], float(loss), arena.flat_grad.clone(), taps))
    (m0, l0, g0, t0), (m1, l1, g1, _) = runs
    assert all(torch.isfinite(m).all() for m in m0) and np.isfinite(l0) and torch.isfinite(g0).all()
    assert all(torch.equal(a, b) for a, b in zip(m0, m1)) and l0 == l1
    worst = _grad_repro(arena, g0, g1)
    print(f"SCD B=16 256^2 T=5 bf16: loss {l0:.5f}, run-to-run gradient rel-L2 (worst parameter) {worst:.2e}")
    assert worst < 5e-5, worst
    assert all(p.grad is not None for _, p in hot_path_named_params(net))
    del runs, m1, g1
    # ---- the same step through the f32 HIP path on the same weights / inputs
    net32 = _build_scd(torch.float32)
    arena32 = ParamArena(hot_path_named_params(net32), torch.device(DEV))
    taps32 = []
    m32, l32 = _scd_step(net32, arena32, pre, post, labels, taps32)
    for i, (a, b) in enumerate(zip(t0, taps32)):
        r = _rel(a, b)
        print(f"SCD stage tap c{i + 1} (3 perception frames): bf16 vs f32 HIP rel-L2 {r:.3e} (bound {TAP_BOUND[i]:.1e})")
        assert r < TAP_BOUND[i], (i, r)
    # semantic argmax maps and the change mask (reference scripts/train_SCD.py:241-246) agree almost everywhere
    for name, a, b in zip(("pre", "post"), m0[:2], m32[:2]):
        agree = (a.argmax(1) == b.argmax(1)).float().mean().item()
        print(f"SCD {name} argmax agreement bf16 vs f32 HIP: {agree:.4f}")
        assert agree > 0.97, (name, agree)
    inter = ((m0[2] > 0.5) & (m32[2] > 0.5)).sum().item()
    union = ((m0[2] > 0.5) | (m32[2] > 0.5)).sum().item()
    print(f"SCD change-mask IoU bf16 vs f32 HIP: {inter / max(union, 1):.4f}; loss {l0:.5f} vs {float(l32):.5f}")
    assert union == 0 or inter / union > 0.95
    assert abs(l0 - float(l32)) < 0.02 * abs(float(l32))
    cos = torch.nn.functional.cosine_similarity(g0.double(), arena32.flat_grad.double(), dim=0).item()
    print(f"SCD flat gradient cosine(bf16, f32 HIP) = {cos:.5f}")
    assert cos > 0.98, cos


# ------------------------------------------------------------------------------------------------- CC
def _build_cc(act_dtype):
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    args = synth.make_cc_args(size=S, vocab_size=501, dropout=0.0)   # dropout off: the comparison is deterministic
    args.act_dtype = act_dtype
    with contextlib.redirect_stdout(io.StringIO()):
        net = Trainer(args)
    sd = synth.synth_state_dict(net, seed=16, branch_gain=0.1)
    sd["decoder.position_encoding.pe"] = net.state_dict()["decoder.position_encoding.pe"].clone()
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    net.decoder.position_encoding.dropout.p = 0.0
    return net


def _cc_step(net, arenas, pre, post, caps, caplens):
    from change3d_amd.model.caption_decoder import packed_cross_entropy
    for a in arenas:
        a.zero_grad()
    feat = net.update_cc(pre, post)
    Bc, Cc, Hc, Wc = feat.shape
    logits = net.decoder.logits_seq_first(feat.permute(2, 3, 0, 1).reshape(Hc * Wc, Bc, Cc), caps)
    loss = packed_cross_entropy(logits, caps, caplens, 501, ignore_index=0)
    loss.backward()
    torch.cuda.synchronize()
    return feat.detach().float(), logits.detach().float(), loss.detach()


def test_cc_bf16_b16_256_step_finite_reproducible_and_tracks_f32():
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.utils import ParamArena, cc_named_params
    pre, post, _ = (t.to(DEV) for t in synth.synth_batch(B, S, seed=0))
    caps, caplens = (t.to(DEV) for t in synth.synth_captions(B, seed=0, vocab_size=501))

    def arenas_of(net):
        enc_named, dec_named = cc_named_params(net)
        return ParamArena(enc_named, torch.device(DEV)), ParamArena(dec_named, torch.device(DEV))

    net = _build_cc(torch.bfloat16)
    arenas = arenas_of(net)
    st0 = {k: v.clone() for k, v in net.state_dict().items()}
    runs = []
    for r in range(2):
        net.load_state_dict(st0)
        feat, logits, loss = _cc_step(net, arenas, pre, post, caps, caplens)
        runs.append((feat.clone(), logits.clone(), float(loss), [a.flat_grad.clone() for a in arenas]))
    (f0, lg0, l0, g0), (f1, lg1, l1, g1) = runs
    assert torch.isfinite(f0).all() and torch.isfinite(lg0).all() and np.isfinite(l0) and all(torch.isfinite(g).all() for g in g0)
    assert f0.shape == (B, 192, 16, 16) and float(f0.std()) > 1e-3
    assert torch.equal(f0, f1) and torch.equal(lg0, lg1) and l0 == l1
    worst = max(_grad_repro(a, x, y) for a, x, y in zip(arenas, g0, g1))
    print(f"CC B=16 256^2 bf16: loss {l0:.5f}, run-to-run gradient rel-L2 (worst parameter) {worst:.2e}")
    assert worst < 5e-5, worst
    del runs, f1, lg1, g1
    net32 = _build_cc(torch.float32)
    arenas32 = arenas_of(net32)
    f32_, lg32, l32 = _cc_step(net32, arenas32, pre, post, caps, caplens)
    r_feat, r_log = _rel(f0, f32_), _rel(lg0, lg32)
    print(f"CC res5 encoder feature (B,192,16,16): bf16 vs f32 HIP rel-L2 {r_feat:.3e} (bound {RES5_BOUND:.1e}); "
          f"logits rel-L2 {r_log:.3e}; loss {l0:.5f} vs {float(l32):.5f}")
    assert r_feat < RES5_BOUND, r_feat
    assert r_log < 2 * RES5_BOUND, r_log
    assert abs(l0 - float(l32)) < 0.02 * abs(float(l32))
    # measured (round 3): encoder 0.974 (70 bf16 blocks between the loss and the stem -- BCD's 55-block path: 0.995), decoder
    # > 0.999 (f32 parameters; only its memory input is bf16-derived)
    for name, ga, gb, lim in zip(("encoder", "decoder"), g0, (a.flat_grad for a in arenas32), (0.95, 0.99)):
        cos = torch.nn.functional.cosine_similarity(ga.double(), gb.double(), dim=0).item()
        print(f"CC flat {name} gradient cosine(bf16, f32 HIP) = {cos:.5f} (bound {lim})")
        assert cos > lim, (name, cos)


# ------------------------------------------------------------------------------------------------- BN statistics
@pytest.mark.parametrize("stage_idx,T", [(2, 5), (3, 5), (4, 3)])
def test_bn_statistics_kernels_vs_f64_recompute_b16(stage_idx, T):
    """B=16 full-size launches the BCD test does not cover: T=5 residual stages (SCD: the two-slot LDS-DMA weight
    gradient and the TT=5 depthwise kernels) and res5 (CC: 432 inner channels on the wide GEMM kernels).  Every
    BatchNorm (mean, rstd) is recomputed in f64 from the activation tensor the stage stored."""
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.x3d import create_x3d, stage_saved_activations
    cin, hw = {2: (24, 128), 3: (48, 64), 4: (96, 32)}[stage_idx]
    net = create_x3d(input_clip_length=T, depth_factor=5.0, act_dtype=torch.bfloat16)
    net.load_state_dict(synth.synth_state_dict(net, seed=21))
    stage = net.blocks[stage_idx].to(DEV).train()
    x = synth.synth_tensor((B, cin, T, hw, hw), 70 + stage_idx).abs().to(DEV)
    xin = x.to(torch.bfloat16).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3).requires_grad_(True)
    y = stage(xin)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    checked = 0
    for bi, rec in enumerate(stage_saved_activations(y)):
        for name in ("a", "b", "c", "sc"):
            act, mr, C = rec.get(name), rec.get("mr_" + name), rec.get("C_" + name)
            if act is None or mr is None:
                continue
            v = act.reshape(-1, act.shape[-1])[:, :C].double()
            mean = v.mean(0)
            var = (v * v).mean(0) - mean * mean
            rstd = 1.0 / torch.sqrt(var + 1e-5)
            Cp = mr.numel() // 2
            e_m = ((mr[:C].double() - mean).abs() / (v.abs().mean(0) + 1e-6)).max().item()
            e_r = ((mr[Cp:Cp + C].double() - rstd).abs() / rstd).max().item()
            assert e_m < 2e-5 and e_r < 1e-4, (stage_idx, bi, name, e_m, e_r)
            checked += 1
    # backward through the same stage: finite input gradient and weight gradients (T=5 / res5 backward kernels)
    y.float().square().mean().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(xin.grad).all() and float(xin.grad.abs().max()) > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in stage.parameters())
    print(f"stage {stage_idx} (T={T}, B={B}): {checked} BatchNorm statistics checked against f64")
    assert checked >= 3 * len(stage.res_blocks)
