"""GPU tests that pin the BENCHMARKED configuration: bf16 activations, B=32 per GPU, 256x256 pairs
(BASELINE.json configs[1]) -- the throughput path, at the size where walk lengths, grid caps, XCD chunk
order and the >8-samples-per-walk code paths differ from the small parity cases.

(a) one full train step at B=32 / 256^2 / bf16, checked through size-independent properties: finite,
    bit-reproducible, BatchNorm statistics kernels against an f64 recomputation from the stored activations,
    per-stage features against the f32 HIP path within a stated bf16 bound;
(b) bf16 vs the oracle at 256^2 / B=2 with a GRADIENT bound: per-parameter relative L2 against the fp64
    oracle, yardstick = the fp32 oracle run with every materialised activation (and, through autograd, the
    gradient flowing back through it) rounded to bf16 -- the same `_noise_check` rule as the f32 tests.

Reference: model/trainer.py:221-241 (update_bcd), scripts/train_BCD.py:200-213 (loss / backward / step)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# bf16 storage bounds, stated: relative L2 distance of the per-stage decoder inputs (c1..c4, the enhanced
# perception frame of stem/res2/res3/res4) between the bf16 and the f32 HIP path on identical weights and inputs.
# bf16 has 8 mantissa bits (2^-9 = 2e-3 relative rounding error per stored tensor); the error random-walks through
# 4 / 19 / 49 / 124 stored tensors behind the four taps and is amplified by train-mode BN (SURVEY BASELINE.md 5:
# CPU bf16-vs-fp32, train BN: max |dp| 0.31; tools/grad_error_report.py: this synthetic-weight network amplifies a
# 2^-24 perturbation 1e5-fold).  Measured on MI355X (round 2, B=32): 3.1e-3 / 1.3e-2 / 6.8e-2 / 4.0e-1 with the
# default (chaotic) synthetic weights -- bounds leave ~2x.  With every residual branch scaled by 0.1 (a
# well-conditioned, trained-network-like stack; `branch_gain`) the same taps must agree ~10x tighter.
BF16_STAGE_BOUND = {1.0: (1.0e-2, 4.0e-2, 1.5e-1, 8.0e-1), 0.1: (1.0e-2, 2.0e-2, 3.0e-2, 5.0e-2)}


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _build(size, act_dtype, seed=16, branch_gain=1.0):
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    args = synth.make_args(size=size)
    args.act_dtype = act_dtype
    net = Trainer(args)
    sd = synth.synth_state_dict(net, seed=seed, mask_margin=0.25, branch_gain=branch_gain)
    net.load_state_dict(sd)
    return net.to(DEV).train(), sd


def _step(net, arena, pre, post, tgt, taps=None):
    """zero_grad, forward (optionally recording the encoder's per-stage taps), loss, backward."""
    from change3d_amd.model.utils import BCEDiceLoss
    arena.zero_grad()
    feats = net.encoder(pre, post)
    if taps is not None:
        taps.extend(f[0].detach().float().clone() for f in feats)
    prob = net.decoder([f[0] for f in feats])
    loss = BCEDiceLoss(prob, tgt)
    loss.backward()
    torch.cuda.synchronize()
    return prob.detach(), loss.detach()


@pytest.mark.parametrize("branch_gain", [1.0, 0.1])
def test_bf16_b32_256_step_finite_reproducible_and_tracks_f32(branch_gain):
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.utils import ParamArena, hot_path_named_params
    B, S = 32, 256
    pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(B, S, seed=0))
    net, sd = _build(S, torch.bfloat16, branch_gain=branch_gain)
    arena = ParamArena(hot_path_named_params(net), torch.device(DEV))
    bufs0 = {k: v.clone() for k, v in net.state_dict().items()}
    runs = []
    for r in range(2):
        net.load_state_dict(bufs0)     # same BN running buffers / parameters for both runs
        taps = []
        prob, loss = _step(net, arena, pre, post, tgt, taps)
        runs.append((prob.clone(), float(loss), arena.flat_grad.clone(), taps))
    (p0, l0, g0, t0), (p1, l1, g1, _) = runs
    # ---- finite
    assert torch.isfinite(p0).all() and np.isfinite(l0) and torch.isfinite(g0).all()
    assert 0.0 < float(p0.min()) and float(p0.max()) < 1.0 and 0.02 < float(p0.std())
    zero_bf16 = {n for n, p in hot_path_named_params(net) if p.grad is None or float(p.grad.abs().max()) == 0.0}
    # ---- reproducible: activations bit-identical, gradients up to f32 leaf-gradient atomics
    assert torch.equal(p0, p1) and l0 == l1
    worst = 0.0
    for n, p, o in zip(arena.names, arena.params, arena.offsets):
        a, b = g0[o:o + p.numel()].double(), g1[o:o + p.numel()].double()
        worst = max(worst, ((a - b).norm() / (a.norm() + 1e-30)).item())
    print(f"B=32 256^2 bf16: loss {l0:.5f}, run-to-run gradient rel-L2 (worst parameter) {worst:.2e}")
    assert worst < 5e-5, worst
    # ---- the same step through the f32 HIP path (the parity-tested path) on the same weights / inputs
    del runs, p1, g1
    net32, _ = _build(S, torch.float32, branch_gain=branch_gain)
    arena32 = ParamArena(hot_path_named_params(net32), torch.device(DEV))
    taps32 = []
    p32, l32 = _step(net32, arena32, pre, post, tgt, taps32)
    # every hot parameter got a gradient -- except where the f32 path has an exactly-zero gradient too (an SE
    # block whose hidden ReLU units are all inactive on this batch passes nothing to its first FC)
    zero_f32 = {n for n, p in hot_path_named_params(net32) if p.grad is None or float(p.grad.abs().max()) == 0.0}
    print(f"parameters with an all-zero gradient: bf16 {sorted(zero_bf16)}  f32 {sorted(zero_f32)}")
    assert zero_bf16 == zero_f32 and all(".norm_b.1.block." in n for n in zero_bf16), (zero_bf16 ^ zero_f32)
    for i, (a, b) in enumerate(zip(t0, taps32)):
        r = ((a - b).norm() / (b.norm() + 1e-30)).item()
        print(f"stage tap c{i + 1} (branch_gain {branch_gain}): bf16 vs f32 HIP rel-L2 {r:.3e} (bound {BF16_STAGE_BOUND[branch_gain][i]:.1e}), "
              f"mean {float(a.mean()):+.4f} vs {float(b.mean()):+.4f}, std {float(a.std()):.4f} vs {float(b.std()):.4f}")
        assert r < BF16_STAGE_BOUND[branch_gain][i], (i, r)
        assert abs(float(a.std()) - float(b.std())) < 0.05 * float(b.std()) + 1e-3
    inter = ((p0 > 0.5) & (p32 > 0.5)).sum().item()
    union = ((p0 > 0.5) | (p32 > 0.5)).sum().item()
    print(f"bf16 vs f32 HIP at B=32: loss {l0:.5f} vs {float(l32):.5f}, mask IoU {inter / max(union, 1):.4f}, "
          f"max|dp| {(p0 - p32).abs().max().item():.3e}")
    assert abs(l0 - float(l32)) < 0.05 * abs(float(l32))
    assert inter / max(union, 1) > (0.8 if branch_gain == 1.0 else 0.95)
    # gradient direction agrees with the f32 path (whole-buffer cosine; per-parameter bounds are test (b))
    cos = torch.nn.functional.cosine_similarity(g0.double(), arena32.flat_grad.double(), dim=0).item()
    print(f"flat gradient cosine(bf16, f32 HIP) = {cos:.5f}")
    assert cos > 0.98, cos


@pytest.mark.parametrize("stage_idx", [1, 2, 3])
def test_bn_statistics_kernels_vs_f64_recompute_b32(stage_idx):
    """B=32 full-size launch of one residual stage in bf16: every BatchNorm (mean, rstd) the statistics
    epilogues + finalize kernels produced is recomputed in f64 from the activation tensor the stage stored
    (the values consumers actually read), per block, and must agree to 2e-5 / 1e-4 relative."""
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.x3d import create_x3d, stage_saved_activations
    B, T = 32, 3
    cin, hw = {1: (24, 256), 2: (24, 128), 3: (48, 64)}[stage_idx]
    net = create_x3d(input_clip_length=3, depth_factor=5.0, act_dtype=torch.bfloat16)
    net.load_state_dict(synth.synth_state_dict(net, seed=21))
    stage = net.blocks[stage_idx].to(DEV).train()
    x = synth.synth_tensor((B, cin, T, hw, hw), 50 + stage_idx).abs().to(DEV)
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
    print(f"stage {stage_idx}: {checked} BatchNorm statistics checked against f64")
    assert checked >= 3 * len(stage.res_blocks)


# ------------------------------------------------------------------------------------- (b)
def _bf16_round_hook(_m, _inp, out):
    return out.to(torch.bfloat16).to(torch.float32)   # differentiable: the gradient is rounded on the way back too


def _oracle_bf16_emulation(om, args, sd):
    """fp32 oracle whose every materialised tensor (conv outputs, stem / block outputs) is rounded to bf16."""
    from oracle import pv
    net = om.Trainer(args)
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv3d, torch.nn.Conv2d, torch.nn.ConvTranspose2d, pv.ResBlock, pv.ResNetBasicStem)):
            if isinstance(m, torch.nn.Conv3d) and m.kernel_size == (1, 1, 1) and m.bias is not None:
                continue   # SE convolutions act on pooled f32 vectors
            m.register_forward_hook(_bf16_round_hook)
    return net.train()


@pytest.mark.parametrize("branch_gain", [0.1, 1.0])
def test_bf16_gradients_vs_oracle_256_b2(branch_gain):
    """branch_gain 1.0 = the default synthetic weights: CHAOTIC (tools/grad_error_report.py: a 2^-24 input rounding
    already moves gradients by ~1e-2, so bf16's 2^-9 storage rounding decorrelates them: ~100 % relative error for
    the HIP path AND for the bf16-emulated oracle alike) -- kept as a distribution-level check only.
    branch_gain 0.1 = the same weights with every residual branch scaled by 0.1 (a trained-network-like stack):
    rounding errors stay in the linear regime, and the per-parameter bf16 gradient error is REQUIRED to be small
    in absolute terms as well as inside the emulated oracle's own error distribution."""
    _need_gpu()
    from oracle import model as om
    from change3d_amd import synthetic as synth
    from change3d_amd.model.utils import BCEDiceLoss, hot_path_named_params
    from test_model_gpu import _noise_check
    S, B = 256, 2
    args = om.make_args(size=S)
    net, sd = _build(S, torch.bfloat16, branch_gain=branch_gain)
    pre, post, tgt = synth.synth_batch(B, S, seed=0)
    ref64 = om.Trainer(args)
    ref64.load_state_dict(sd)
    ref64 = ref64.double().train()
    p64 = ref64.update_bcd(pre.double(), post.double())
    l64 = om.bce_dice_loss(p64, tgt.double())
    l64.backward()
    refb = _oracle_bf16_emulation(om, args, sd)
    pb = refb.update_bcd(pre, post)
    lb = om.bce_dice_loss(pb, tgt)
    lb.backward()
    pd = net.update_bcd(pre.to(DEV), post.to(DEV))
    ld = BCEDiceLoss(pd, tgt.to(DEV))
    ld.backward()
    torch.cuda.synchronize()
    names = [n for n, _ in hot_path_named_params(net)]
    g_hip = {n: p.grad for n, p in hot_path_named_params(net)}
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    gb = {n: p.grad for n, p in refb.named_parameters() if p.grad is not None}
    e_hip = np.array([(g_hip[n].double().cpu() - g64[n]).norm().item() / (g64[n].norm().item() + 1e-30) for n in names])
    e_ref = np.array([(gb[n].double() - g64[n]).norm().item() / (g64[n].norm().item() + 1e-30) for n in names])
    ep_hip = (pd.detach().cpu().double() - p64.detach()).abs().max().item()
    ep_ref = (pb.detach().double() - p64.detach()).abs().max().item()
    print(f"bf16 256^2 B=2 branch_gain {branch_gain}: max|p - p_fp64| hip {ep_hip:.3e} / bf16-emulated oracle {ep_ref:.3e};  "
          f"loss hip {ld.item():.5f} oracle-bf16 {lb.item():.5f} fp64 {l64.item():.5f}")
    _noise_check(names, e_hip, e_ref, f"bf16 grad rel-L2 (branch_gain {branch_gain})")
    assert ep_hip <= 4.0 * ep_ref + 1e-3
    assert abs(ld.item() - l64.item()) <= 4.0 * abs(lb.item() - l64.item()) + 1e-3
    if branch_gain < 1.0:
        # linear regime: bounds in absolute terms (bf16 keeps 8 bits: 2^-9 = 2e-3 per stored tensor, accumulated
        # over the up to ~250 stored tensors between a parameter and the loss)
        print(f"   linear regime: hip median {np.median(e_hip):.2e} p90 {np.percentile(e_hip, 90):.2e} max {e_hip.max():.2e}")
        # measured (round 2): HIP median 4.1e-2 / p90 6.9e-2, emulated oracle median 3.9e-2; max|dp| 1.6e-2 vs 1.3e-2
        assert np.median(e_hip) < 8e-2 and np.percentile(e_hip, 90) < 1.5e-1 and ep_hip < 5e-2
        inter = ((pd.detach().cpu() > 0.5) & (p64.detach() > 0.5)).sum().item()
        union = ((pd.detach().cpu() > 0.5) | (p64.detach() > 0.5)).sum().item()
        print(f"   change-mask IoU bf16 HIP vs fp64 oracle: {inter / max(union, 1):.4f}")
        assert union == 0 or inter / union > 0.97
