"""GPU parity tests, module level and end to end: the HIP path behind the reference's module
surface (create_x3d().blocks[i], Encoder.enhance, ChangeDecoder, Trainer.update_bcd,
BCEDiceLoss, Adam) against the CPU oracle on the same seeded weights/inputs, and against
the committed golden vectors produced by the REAL reference (tests/golden/*.npz,
oracle/gen_golden.py).

Tolerance rule (f32 path).  Module-level tests use fixed tolerances (1e-4 abs on outputs, 1-2e-3
relative L2 on gradients).  For the 55-block end-to-end network the fp32 CPU reference is itself
7.7e-5 (probabilities) / up to 3 % (per-parameter gradient L2, median 0.7 %) away from an fp64
evaluation of the same modules, so end-to-end quantities are judged against fp64:
    |hip - fp64|  <=  K_NOISE * |reference_fp32 - fp64|  (+ a small floor),   K_NOISE = 4,
and change masks must be bit-exact wherever the fp64 probability is further than that
tolerance from the 0.5 threshold."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).norm().item() / (b.norm().item() + 1e-20)


def load_matching(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def randomize(module, seed):
    from oracle import synth
    module.load_state_dict(synth.synth_state_dict(module, seed=seed))


@pytest.mark.parametrize("cfg", [dict(cin=24, cinner=54, cout=24), dict(cin=24, cinner=108, cout=48),
                                 dict(cin=48, cinner=216, cout=96), dict(cin=96, cinner=432, cout=192)])
def test_res_stage_fwd_bwd(cfg):
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.x3d import X3DResStage
    depth, B, T, H, W = 3, 2, 3, 16, 16
    ref = om.build_stage(depth, cfg["cin"], cfg["cinner"], cfg["cout"], (1, 2, 2))
    randomize(ref, 3)
    mine = X3DResStage(depth, cfg["cin"], cfg["cinner"], cfg["cout"], 2, 0.0625, torch.float32)
    load_matching(mine, ref)
    mine = mine.to(DEV).train()
    ref.train()
    x = synth.synth_tensor((B, cfg["cin"], T, H, W), 5).abs()
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    go = synth.synth_tensor(tuple(yr.shape), 6)
    yr.backward(go)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(xd)
    yd.backward(go.to(DEV))
    torch.cuda.synchronize()
    assert yd.shape == yr.shape
    assert (yd.cpu() - yr).abs().max().item() < 2e-4 * max(1.0, yr.abs().max().item()), (yd.cpu() - yr).abs().max()
    assert rel(xd.grad, xr.grad) < 2e-3, ("dx", rel(xd.grad, xr.grad))
    pr = dict(ref.named_parameters())
    worst = []
    for n, p in mine.named_parameters():
        assert p.grad is not None, n
        r = rel(p.grad, pr[n].grad)
        if r > 2e-3 and (p.grad.cpu() - pr[n].grad).abs().max().item() > 1e-5:
            worst.append((n, r))
    assert not worst, worst
    br = dict(ref.named_buffers())
    for n, b in mine.named_buffers():
        if b.dtype == torch.int64:
            assert int(b) == int(br[n]), n
        else:
            assert torch.allclose(b.cpu(), br[n], rtol=1e-4, atol=1e-5), n
    # eval mode
    ref.eval(); mine.eval()
    with torch.no_grad():
        ye, yde = ref(x), mine(x.to(DEV))
    assert (yde.cpu() - ye).abs().max().item() < 2e-4 * max(1.0, ye.abs().max().item())


@pytest.mark.parametrize("T,dtype,shape", [(3, torch.float32, (2, 24, 40)), (5, torch.float32, (2, 24, 40)),
                                           (4, torch.float32, (3, 17, 70)), (3, torch.bfloat16, (2, 24, 40)),
                                           (5, torch.bfloat16, (1, 40, 33))])
def test_stem_fwd_bwd(T, dtype, shape):
    """Stem (MFMA kernels, csrc/stem_mfma.hip) vs the oracle: ragged tiles in both directions, T = 3 / 4 / 5, both
    storage types (the arithmetic is f32 on both: only u / dv are rounded on the bf16 path)."""
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.x3d import X3DStem
    B, H, W = shape
    bf = dtype == torch.bfloat16
    tol_y, tol_g = (1.5e-2, 2e-2) if bf else (1e-4, 2e-3)
    ref = om.build_stem(3, 24)
    randomize(ref, 7)
    mine = X3DStem(3, 24, (5, 3, 3), (1, 1, 1), dtype)
    load_matching(mine, ref)
    mine = mine.to(DEV).train()
    ref.train()
    x = synth.synth_tensor((B, 3, T, H, W), 8)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    go = synth.synth_tensor(tuple(yr.shape), 9)
    yr.backward(go)
    xd = x.to(DEV).requires_grad_(True)
    yd = mine(xd)
    yd.backward(go.to(DEV).to(yd.dtype))
    assert (yd.float().cpu() - yr).abs().max().item() < tol_y * max(1.0, yr.abs().max().item() if bf else 1.0)
    assert rel(xd.grad, xr.grad) < tol_g
    pr = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        assert rel(p.grad, pr[n].grad) < tol_g, (n, rel(p.grad, pr[n].grad))
    # restricted input gradient (perception frames only)
    mine.grad_frames = (1, T - 2)
    xd2 = x.to(DEV).requires_grad_(True)
    mine.zero_grad()
    yd2 = mine(xd2)
    yd2.backward(go.to(DEV).to(yd2.dtype))
    assert rel(xd2.grad[:, :, 1:T - 1], xr.grad[:, :, 1:T - 1]) < tol_g
    assert xd2.grad[:, :, 0].abs().max().item() == 0


@pytest.mark.parametrize("K", [1, 3])
def test_enhance_fwd_bwd(K):
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.trainer import _EnhanceFn
    B, C, T, H, W = 2, 48, K + 2, 16, 16
    args = om.make_args(num_perception_frame=K, size=16)
    enc = om.Encoder.__new__(om.Encoder)
    torch.nn.Module.__init__(enc)
    enc.args = args
    fc = torch.nn.Sequential(torch.nn.Conv2d(C, C, 1, bias=False), torch.nn.ReLU())
    x = synth.synth_tensor((B, C, T, H, W), 10).abs()
    x[:, :, 0, :2] = x[:, :, K + 1, :2]  # exact ties: |pre-post| = 0 -> sign 0
    xr = x.clone().requires_grad_(True)
    yr = om.Encoder.enhance(enc, xr, fc)
    go = synth.synth_tensor(tuple(yr.shape), 11)
    yr.backward(go)
    w = fc[0].weight.detach().clone().to(DEV).requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    yd = _EnhanceFn.apply(xd, w, K + 1)
    yd.backward(go.to(DEV))
    assert (yd.cpu() - yr).abs().max().item() < 1e-4
    assert rel(xd.grad, xr.grad) < 1e-3
    assert rel(w.grad, fc[0].weight.grad) < 1e-3


@pytest.mark.parametrize("has_sigmoid,nc", [(True, 1), (False, 7)])
def test_change_decoder_fwd_bwd(has_sigmoid, nc):
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.change_decoder import ChangeDecoder
    args = om.make_args(num_class=nc)
    ref = om.ChangeDecoder(args, in_dim=[24, 24, 48, 96], has_sigmoid=has_sigmoid)
    randomize(ref, 12)
    mine = ChangeDecoder(args, in_dim=[24, 24, 48, 96], has_sigmoid=has_sigmoid)
    load_matching(mine, ref)
    mine = mine.to(DEV)
    B, T, S = 2, 3, 64
    dims = [(24, S), (24, S // 2), (48, S // 4), (96, S // 8)]
    full = [synth.synth_tensor((B, c, T, s, s), 20 + i, 0.5) for i, (c, s) in enumerate(dims)]
    fr = [f.clone().requires_grad_(True) for f in full]
    out_r = ref([f[:, :, 1] for f in fr])
    go = synth.synth_tensor(tuple(out_r.shape), 30)
    out_r.backward(go)
    # device: channels_last_3d storage so that x[:, :, 1] is an in-place NHWC frame view
    fd = [f.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True) for f in full]
    out_d = mine([f[:, :, 1] for f in fd])
    out_d.backward(go.to(DEV))
    assert (out_d.cpu() - out_r).abs().max().item() < 1e-4 * max(1.0, out_r.abs().max().item())
    for i in range(4):
        assert rel(fd[i].grad, fr[i].grad) < 1e-3, (i, rel(fd[i].grad, fr[i].grad))
    pr = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        assert rel(p.grad, pr[n].grad) < 1e-3, (n, rel(p.grad, pr[n].grad))


def _build_pair(size, act_dtype=torch.float32, branch_gain=None):
    from oracle import model as om, synth
    from change3d_amd.model.trainer import Trainer
    args = om.make_args(size=size)
    ref = om.Trainer(args)
    kw = {} if branch_gain is None else {"branch_gain": branch_gain}
    sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25, **kw)
    ref.load_state_dict(sd)
    args2 = om.make_args(size=size)
    args2.act_dtype = act_dtype
    mine = Trainer(args2)
    mine.load_state_dict(sd)
    return ref, mine.to(DEV), sd


def test_trainer_state_dict_keys_match_oracle():
    _need_gpu()
    ref, mine, _ = _build_pair(64)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    assert all(a.shape == b.shape for a, b in zip(ref.state_dict().values(), mine.state_dict().values()))


K_NOISE = 4.0  # HIP error vs fp64 may be at most this multiple of the reference's own fp32 error vs fp64


def _grad_check(names, g_hip, g32, g64):
    """Per-parameter relative L2 error against the fp64 evaluation, bounded by the fp32
    reference's own error (random-weight train-mode BN networks are ill-conditioned: the fp32
    CPU reference itself is up to ~3 % away from fp64 on some parameters, median ~0.7 %)."""
    e_hip = np.array([(g_hip[n].double().cpu() - g64[n]).norm().item() / (g64[n].norm().item() + 1e-30) for n in names])
    e_ref = np.array([(g32[n].double() - g64[n]).norm().item() / (g64[n].norm().item() + 1e-30) for n in names])
    _noise_check(names, e_hip, e_ref, "grad rel-L2")


def _noise_check(names, e_hip, e_ref, what):
    """The HIP error distribution must look like the fp32 reference's own error distribution:
    same median (x1.5), bounded maximum (x K_NOISE), and at most 2 % of the parameters outside
    K_NOISE x max(their own reference error, the reference's 90th-percentile error)."""
    lim = np.maximum(np.maximum(K_NOISE * e_ref, K_NOISE * np.percentile(e_ref, 90)), 2e-3)
    out = np.nonzero(e_hip > lim)[0]
    print(f"{what} vs fp64: hip median {np.median(e_hip):.2e} max {e_hip.max():.2e} | "
          f"fp32 reference median {np.median(e_ref):.2e} max {e_ref.max():.2e} | outliers {len(out)}/{len(names)}")
    for i in out[:40]:
        print(f"   outlier {names[i]}: hip {e_hip[i]:.2e} reference {e_ref[i]:.2e}")
    assert np.median(e_hip) <= 1.5 * np.median(e_ref) + 1e-4, (np.median(e_hip), np.median(e_ref))
    assert e_hip.max() <= K_NOISE * e_ref.max() + 2e-3, (str(names[int(e_hip.argmax())]), e_hip.max(), e_ref.max())
    assert len(out) <= 0.02 * len(names), [(str(names[i]), float(e_hip[i]), float(e_ref[i])) for i in out[:10]]


def test_e2e_bcd_vs_oracle_size64():
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.utils import BCEDiceLoss, hot_path_named_params
    ref, mine, sd = _build_pair(64)
    ref64 = om.Trainer(om.make_args(size=64))
    ref64.load_state_dict(sd)
    ref64 = ref64.double()
    pre, post, tgt = synth.synth_batch(2, 64, seed=0)
    ref.train(); mine.train(); ref64.train()
    pr = ref.update_bcd(pre, post)
    lr_ = om.bce_dice_loss(pr, tgt)
    lr_.backward()
    p64 = ref64.update_bcd(pre.double(), post.double())
    l64 = om.bce_dice_loss(p64, tgt.double())
    l64.backward()
    pd = mine.update_bcd(pre.to(DEV), post.to(DEV))
    ld = BCEDiceLoss(pd, tgt.to(DEV))
    ld.backward()
    torch.cuda.synchronize()
    p64 = p64.detach()
    e_hip = (pd.detach().cpu().double() - p64).abs().max().item()
    e_ref = (pr.detach().double() - p64).abs().max().item()
    print(f"max |p - p_fp64|: hip {e_hip:.3e}  fp32 reference {e_ref:.3e}   hip vs fp32 reference "
          f"{(pd.detach().cpu() - pr.detach()).abs().max().item():.3e}")
    tol_p = K_NOISE * e_ref + 1e-6
    assert e_hip <= tol_p, (e_hip, e_ref)
    assert abs(ld.item() - l64.item()) <= K_NOISE * abs(lr_.item() - l64.item()) + 1e-5
    # change mask: bit-exact wherever the fp64 probability is further than the tolerance from 0.5
    safe = (p64 - 0.5).abs() > tol_p
    assert (((pd.detach().cpu() > 0.5) != (p64 > 0.5)) & safe).sum().item() == 0
    names = [n for n, _ in hot_path_named_params(mine)]
    g_hip = {n: p.grad for n, p in hot_path_named_params(mine)}
    for n in names:
        assert g_hip[n] is not None, n
    g32 = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    assert set(names) == set(g32.keys())
    _grad_check(names, g_hip, g32, g64)
    for n, p in mine.named_parameters():
        if n.startswith(("encoder.x3d.blocks.4.", "encoder.x3d.blocks.5.")):
            assert p.grad is None, n


def _conditioned_case(task, wseed, tol=1e-4):
    """One (task, weight seed) case of the conditioned-weights comparison: HIP f32 vs the fp32 oracle, EVERY gradient tensor
    to `tol` relative L2 -- kink-robust (oracle/kinks.py).  A ReLU pre-activation within f32 noise of zero takes either side
    depending on who sums in which order (torch-CPU itself changes sides with its thread count), and one flipped unit moves the
    ~100 tensors upstream of it by 1e-4 .. 1e-3.  So: the oracle runs at two thread counts and every tensor is judged against
    the closer one; if that already meets `tol`, done.  Otherwise the f64 oracle's pre-activations name the units AT RISK
    (|pre| < 6 sigma of the f32 oracle's own noise at that ReLU call), one f64 run per unit gives what flipping it does to every
    gradient, and a least-squares fit says which of them the HIP path took on the other side; those (printed) are granted,
    and every tensor must meet `tol` against oracle + their deltas.  A flip of a unit that is not at risk, or an error that is
    not a sum of such deltas, fails as before.  Returns ({parameter: gradient rel-L2}, [granted units])."""
    from oracle import kinks, model as om, synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss, hot_path_named_params
    size, batch = 64, 2
    mk = (lambda: om.make_args(size=size)) if task == "bcd" else \
        (lambda: om.make_args(num_perception_frame=3, size=size, dataset="SECOND", num_class=7))
    sd = synth.synth_state_dict(om.Trainer(mk()), seed=wseed, mask_margin=0.25, branch_gain=0.1)
    pre, post, tgt = synth.synth_batch(batch, size, seed=0)
    labels = synth.synth_scd_labels(batch, size, seed=0) if task == "scd" else None
    mine = Trainer(mk())
    mine.load_state_dict(sd)
    mine = mine.to(DEV).train()
    if task == "bcd":
        outs_d = [mine.update_bcd(pre.to(DEV), post.to(DEV))]
        BCEDiceLoss(outs_d[0], tgt.to(DEV)).backward()
    else:
        from change3d_amd.model.utils import ChangeSimilarity, CrossEntropyLoss2d
        from change3d_amd.scripts.train_SCD import scd_loss
        outs_d = list(mine.update_scd(pre.to(DEV), post.to(DEV)))
        scd_loss(CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity(), outs_d, labels.to(DEV))[0].backward()
    torch.cuda.synchronize()
    g_hip = {n: p.grad.detach().cpu() for n, p in hot_path_named_params(mine)}

    def oracle_run(dtype):
        ref = om.Trainer(mk())
        ref.load_state_dict(sd)
        ref = (ref.double() if dtype == torch.float64 else ref).train()

        def run():
            ref.zero_grad(set_to_none=True)
            a, b = pre.to(dtype), post.to(dtype)
            if task == "bcd":
                outs = [ref.update_bcd(a, b)]
                om.bce_dice_loss(outs[0], tgt.to(dtype)).backward()
            else:
                outs = list(ref.update_scd(a, b))
                om.scd_loss(*outs, labels).backward()
            return outs
        return ref, run

    def check(outs_r):
        for od, orr in zip(outs_d, outs_r):
            assert (od.detach().cpu() - orr.detach()).abs().max().item() < 1e-5 * max(1.0, orr.detach().abs().max().item())

    print(f"{task} seed {wseed}:")
    errs, granted = kinks.strict_compare(g_hip, oracle_run, tol=tol, check_outputs=check)
    if max(errs.values()) >= tol and os.path.isdir("gpurun_out"):   # keep the evidence: the fit can be replayed on a CPU box
        torch.save(g_hip, f"gpurun_out/kink_fail_{task}_{wseed}.pt")
    return errs, granted


def _summ(task, wseed, errs, granted=()):
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    over = [n for n, e in errs.items() if e >= 1e-4]
    med = sorted(errs.values())[len(errs) // 2]
    print(f"{task} conditioned (weight seed {wseed}): worst per-parameter gradient rel-L2 {[(n, f'{e:.1e}') for n, e in worst]}; "
          f"median {med:.1e}; {len(over)}/{len(errs)} tensors above 1e-4; ReLU units granted the other side: "
          f"{[(u[0], u[1], f'{u[3]:.2f} sigma') for u in granted]}")
    return worst[0][1], med, len(over)


def test_e2e_vs_oracle_conditioned_weights_every_gradient_bcd():
    """The default synthetic weights are chaotic (tools/grad_error_report.py: one input rounding moves gradients by
    ~1e-2), which is why the tests around this one judge errors against the fp32 reference's own distance from fp64.
    With every residual branch scaled by 0.1 (`branch_gain`: a trained-network-like, well-conditioned stack) fp32
    rounding stays in the linear regime and the HIP f32 path is compared DIRECTLY with the fp32 oracle: outputs to
    1e-5, EVERY parameter gradient to 1e-4 relative L2 (measured: worst 2e-5), on four weight seeds, kink-robustly
    (`_conditioned_case`): at most a handful of at-risk ReLU units may be granted the other side, and they are printed."""
    _need_gpu()
    n_granted = []
    for wseed in (16, 23, 26, 27):
        errs, granted = _conditioned_case("bcd", wseed)
        worst, _, _ = _summ("bcd", wseed, errs, granted)
        assert worst < 1e-4 and len(granted) <= 6, (wseed, worst, granted)
        assert all(u[3] < 6.0 for u in granted), ("a granted unit must lie within 6 sigma of zero", wseed, granted)
        n_granted.append(len(granted))
    # canary: granting must stay the exception -- on at least one of the four seeds the strict bound holds with NO unit granted
    assert min(n_granted) == 0, n_granted


def test_e2e_vs_oracle_conditioned_weights_every_gradient_scd():
    """SCD (T=5, three decoders): the same STRICT bound -- every one of the 490 gradient tensors within 1e-4 of the fp32
    oracle -- on weight seeds 23, 26, 27 AND the default seed 16, nothing waived.  There are ~120 ReLU layers over 5/3 as many
    elements as BCD; until round 5 the strict bound could only be asserted on seeds scanned to be kink-free under the kernels'
    exact roundings (tools/scan_scd_seed.py: 47 of 54 scanned seeds had a unit that flipped), which let three seeds veto any
    kernel change that re-associates a sum (the f32 depthwise forward had to keep a slower LDS layout than the bf16 one).
    Now a flip is identified and granted by name (`_conditioned_case`, oracle/kinks.py): seed 16 -- the kinked one -- meets the
    strict bound as well, and a regression still fails on every seed (it is not a sum of at-risk deltas)."""
    _need_gpu()
    n_granted = []
    for wseed in (23, 26, 27, 16):
        errs, granted = _conditioned_case("scd", wseed)
        worst, med, n_over = _summ("scd", wseed, errs, granted)
        assert worst < 1e-4 and med < 2e-5 and len(granted) <= 8, (wseed, worst, med, granted)
        assert all(u[3] < 6.0 for u in granted), ("a granted unit must lie within 6 sigma of zero", wseed, granted)
        n_granted.append(len(granted))
    assert min(n_granted) == 0, n_granted   # canary (see the BCD case)


@pytest.mark.parametrize("T,dtype,shape", [(3, torch.float32, (2, 64, 64)), (5, torch.float32, (2, 40, 72)),
                                           (3, torch.bfloat16, (3, 128, 128)), (4, torch.bfloat16, (1, 17, 70)),
                                           (5, torch.bfloat16, (2, 64, 64))])
def test_stem_mfma_kernels_equal_the_scalar_kernels(T, dtype, shape):
    """csrc/stem_mfma.hip against csrc/stem.hip through the same C entry points (c3d_set_option C3D_OPT_STEM_MFMA):
    u, dv and the input gradient are BIT-identical (v_mfma_f32_16x16x4_f32 accumulates k in order, like the FMA chain);
    sums that go through atomics or a different partial-sum grouping (statistics, weight gradients) agree to 2e-6."""
    _need_gpu()
    from change3d_amd import ops
    B, H, W = shape
    dt = ops.dt_code(dtype)

    def run(mfma):
        ops.set_option(ops.OPT_STEM_MFMA, 1 if mfma else 0)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, 3, T, H, W, generator=g).to(DEV)
        w_t = (torch.randn(24, 3, 1, 3, 3, generator=g) * 0.3).to(DEV)
        w_xy = (torch.randn(24, 1, 5, 1, 1, generator=g) * 0.5).to(DEV)
        u = torch.zeros(B, T, H, W, 24, dtype=dtype, device=DEV)
        sums = torch.zeros(48, dtype=torch.float64, device=DEV)
        ops.stem_fwd(x, w_t, w_xy, u, sums, B, T, H, W, dt)
        g0 = torch.randn(B, T, H, W, 24, generator=g).to(DEV).to(dtype)
        coef = torch.randn(72, generator=g).to(DEV)
        dv = torch.zeros_like(u)
        dw_xy = torch.zeros(24, 5, device=DEV)
        ops.stem_bwd_dv(x, w_t, w_xy, g0, u, coef, dv, dw_xy, B, T, H, W, dt)
        dw_t = torch.zeros(24, 27, device=DEV)
        dP = torch.zeros(B, 3, T, H, W, device=DEV)
        ops.stem_bwd_wx(x, w_t, dv, dw_t, dP, B, T, H, W, 0, T, True, dt)
        dPs = torch.zeros(3, T - 2, H, W, device=DEV)
        dw_t2 = torch.zeros(24, 27, device=DEV)
        ops.stem_bwd_wx(x, w_t, dv, dw_t2, dPs, B, T, H, W, 1, T - 2, False, dt)
        torch.cuda.synchronize()
        return dict(u=u, dv=dv, dP=dP), dict(sums=sums, dw_xy=dw_xy, dw_t=dw_t, dw_t2=dw_t2, dPs=dPs)

    try:
        (ea, sa), (eb, sb) = run(True), run(False)
    finally:
        ops.set_option(ops.OPT_STEM_MFMA, 2)
    for k in ea:
        assert torch.equal(ea[k], eb[k]), k
    for k in sa:
        assert (sa[k] - sb[k]).abs().max().item() <= 2e-6 * sb[k].abs().max().item(), k
    assert eb["u"].float().abs().max().item() > 0.1 and eb["dP"].abs().max().item() > 0.1


@pytest.mark.parametrize("T,shape", [(3, (2, 40, 72)), (3, (3, 17, 70)), (3, (5, 64, 64)), (4, (2, 16, 48)), (5, (2, 24, 40)), (3, (40, 16, 32))])
def test_stem_bwd_wx_on_the_bf16_matrix_cores(T, shape):
    """C3D_OPT_STEM_MFMA = 2 (default): c3d_stem_bwd_wx of bf16 storage multiplies on the bf16 matrix cores (x and w_t rounded to
    bf16 for the products; reference model/x3d.py:84-95 conv_xy backward).  Against the f32 matrix-core kernel on the same
    bf16 dv: the rounding of one operand of every product, relative to the largest entry."""
    _need_gpu()
    from change3d_amd import ops
    B, H, W = shape
    dt = ops.dt_code(torch.bfloat16)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, T, H, W, generator=g).to(DEV)
    w_t = (torch.randn(24, 3, 1, 3, 3, generator=g) * 0.3).to(DEV)
    dv = torch.randn(B, T, H, W, 24, generator=g).to(DEV).to(torch.bfloat16)

    def run(opt):
        ops.set_option(ops.OPT_STEM_MFMA, opt)
        dw_t = torch.ones(24, 27, device=DEV)                   # accumulate semantics
        dPs = torch.ones(3, T - 2, H, W, device=DEV)
        ops.stem_bwd_wx(x, w_t, dv, dw_t, dPs, B, T, H, W, 1, T - 2, False, dt)
        dw_t2 = torch.zeros(24, 27, device=DEV)
        dP = torch.zeros(B, 3, T, H, W, device=DEV)
        ops.stem_bwd_wx(x, w_t, dv, dw_t2, dP, B, T, H, W, 1, T - 2, True, dt)
        dw_t3 = torch.zeros(24, 27, device=DEV)
        ops.stem_bwd_wx(x, w_t, dv, dw_t3, None, B, T, H, W, 0, 0, False, dt)
        torch.cuda.synchronize()
        return dict(dw_t=dw_t, dPs=dPs, dw_t2=dw_t2, dP=dP, dw_t3=dw_t3)

    try:
        a, b = run(2), run(1)
    finally:
        ops.set_option(ops.OPT_STEM_MFMA, 2)
    for k in a:
        err = (a[k] - b[k]).abs().max().item() / b[k].abs().max().item()
        assert err < 6e-3, (k, err)
    assert torch.equal(a["dP"][:, :, 0], torch.zeros_like(a["dP"][:, :, 0]))   # only the perception frames are written
    assert (a["dw_t2"] - a["dw_t3"]).abs().max().item() <= 2e-6 * a["dw_t3"].abs().max().item()
    # and against an f64 convolution of the same tensors
    xd, wd = x.double().cpu(), w_t.double().cpu()
    dvd = dv.double().cpu().permute(0, 4, 1, 2, 3)             # [B][24][T][H][W]
    xd.requires_grad_(True); wd.requires_grad_(True)
    v = torch.nn.functional.conv3d(xd, wd, padding=(0, 1, 1))
    (v * dvd).sum().backward()
    ref_dw = wd.grad.reshape(24, 27).float()
    ref_dx = xd.grad[:, :, 1:T - 1].sum(0).float()              # [3][T-2][H][W]
    e_dw = (a["dw_t3"].cpu() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
    e_dx = (a["dPs"].cpu() - 1.0 - ref_dx).abs().max().item() / ref_dx.abs().max().item()
    assert e_dw < 6e-3 and e_dx < 6e-3, (e_dw, e_dx)


@pytest.mark.parametrize("size", [64, 256])
def test_e2e_bcd_vs_reference_golden(size, golden_dir):
    """Against vectors produced by the real reference in fp32 (and its fp64 evaluation as the
    noise yardstick) — oracle/gen_golden.py."""
    _need_gpu()
    from oracle import synth
    from change3d_amd.model.utils import BCEDiceLoss, FusedAdam, ParamArena, adjust_learning_rate, hot_path_named_params
    from change3d_amd.utils.metric_tool import ConfuseMatrixMeter
    G = np.load(os.path.join(golden_dir, f"bcd_s{size}_b2.npz"))
    _, mine, sd = _build_pair(size)
    pre, post, tgt = synth.synth_batch(2, size, seed=0)
    pre, post, tgt = pre.to(DEV), post.to(DEV), tgt.to(DEV)
    stride = max(1, size // 32)

    def lattice(t):
        return t.detach()[:, :, ::stride, ::stride].cpu().double().numpy()

    # ---- eval mode (BN running statistics := batch statistics of one momentum-1.0 train pass)
    mine.train()
    bns = [m for m in mine.modules() if isinstance(m, torch.nn.BatchNorm3d)]
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        mine.update_bcd(pre, post)
    for m in bns:
        m.momentum = 0.1
    mine.eval()
    with torch.no_grad():
        pe = mine.update_bcd(pre, post)
    mine.load_state_dict(sd)  # back to the synthetic running statistics for the training part
    e_ref = np.abs(G["eval_prob_lattice"] - G["eval_prob_lattice_f64"]).max()
    e_hip = np.abs(lattice(pe) - G["eval_prob_lattice_f64"]).max()
    print(f"eval  max|p - p_fp64| on lattice: hip {e_hip:.3e}  fp32 reference {e_ref:.3e}")
    tol_e = K_NOISE * e_ref + 1e-6
    assert e_hip <= tol_e
    assert 0.05 < float(pe.mean()) < 0.95 and float(pe.std()) > 0.05  # the eval fixture is not saturated
    safe = np.abs(G["eval_prob_lattice_f64"] - 0.5) > tol_e
    assert (((lattice(pe) > 0.5) != (G["eval_prob_lattice_f64"] > 0.5)) & safe).sum() == 0
    bits = np.unpackbits(G["eval_mask_bits"])[:pe.numel()].reshape(pe.shape).astype(bool)
    flips = ((pe.cpu().numpy() > 0.5) != bits).sum()
    assert flips <= K_NOISE * int(G["eval_band"]) + 2, (flips, int(G["eval_band"]))
    # ---- N training steps, reference schedule
    mine.train()
    args = mine.args
    args.lr_mode, args.lr, args.max_epochs, args.step_loss = "poly", float(G["base_lr"]), 1, 100
    arena = ParamArena(hot_path_named_params(mine), torch.device(DEV))
    opt = FusedAdam(arena, lr=args.lr)
    meter = ConfuseMatrixMeter(2)
    losses = []
    for it in range(int(G["meta"][4])):
        lr = adjust_learning_rate(args, opt, 0, it, int(G["max_iter"]))
        assert abs(lr - G["lr_curve"][it]) < 1e-15
        prob = mine.update_bcd(pre, post)
        loss = BCEDiceLoss(prob, tgt)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            e_ref = np.abs(G["train_prob_lattice"] - G["train_prob_lattice_f64"]).max()
            e_hip = np.abs(lattice(prob) - G["train_prob_lattice_f64"]).max()
            print(f"train max|p - p_fp64| on lattice: hip {e_hip:.3e}  fp32 reference {e_ref:.3e}")
            tol_p = K_NOISE * e_ref + 1e-6
            assert e_hip <= tol_p
            safe = np.abs(G["train_prob_lattice_f64"] - 0.5) > tol_p
            assert (((lattice(prob) > 0.5) != (G["train_prob_lattice_f64"] > 0.5)) & safe).sum() == 0
            tb = np.unpackbits(G["train_mask_bits"])[:prob.numel()].reshape(prob.shape).astype(bool)
            assert ((prob.detach().cpu().numpy() > 0.5) != tb).sum() <= K_NOISE * int(G["train_band"]) + 2
            named = dict(mine.named_parameters())
            gn = np.array([named[str(n)].grad.double().norm().item() for n in G["grad_names"]])
            n64 = G["grad_norms_f64"]
            eh = np.abs(gn - n64) / (n64 + 1e-30)
            er = np.abs(G["grad_norms"] - n64) / (n64 + 1e-30)
            _noise_check(G["grad_names"], eh, er, "grad-norm rel err")
            unused = sum(p.numel() for n, p in mine.named_parameters() if p.grad is None)
            assert unused == int(G["unused_param_count"])
        opt.step()
        meter.update_cm_device(prob, tgt)
        losses.append(loss.item())
    losses = np.array(losses)
    l_tol = K_NOISE * np.abs(G["loss_curve"] - G["loss_curve_f64"]).max() + 1e-4
    print(f"loss curve hip {losses} ref32 {G['loss_curve']} ref64 {G['loss_curve_f64']}")
    assert (np.abs(losses - G["loss_curve_f64"]) <= l_tol).all(), (losses, G["loss_curve"], G["loss_curve_f64"])
    meter.sync()
    assert np.abs(meter.sum - G["cm_total"]).sum() <= 1e-3 * G["cm_total"].sum()
    fin = mine.state_dict()
    l2 = np.array([fin[str(n)].double().norm().item() for n in G["grad_names"]])
    assert (np.abs(l2 - G["final_param_l2"]) / (G["final_param_l2"] + 1e-12)).max() < 1e-4
    rm = [v.double().sum().item() for k, v in fin.items() if k.endswith("running_mean") and ".blocks.4." not in k and ".blocks.5." not in k]
    assert np.allclose(np.array(rm), G["final_running_mean_sums"], rtol=2e-3, atol=1e-3)
    nbt = np.array([int(v) for k, v in fin.items() if k.endswith("num_batches_tracked")])
    assert (nbt == G["final_nbt"]).all()


def _build_pair_k(size, k, num_class, act_dtype=torch.float32):
    from oracle import model as om, synth
    from change3d_amd.model.trainer import Trainer
    ref = om.Trainer(om.make_args(num_perception_frame=k, size=size, dataset="SECOND", num_class=num_class))
    sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25)
    ref.load_state_dict(sd)
    args2 = om.make_args(num_perception_frame=k, size=size, dataset="SECOND", num_class=num_class)
    args2.act_dtype = act_dtype
    mine = Trainer(args2)
    mine.load_state_dict(sd)
    return ref, mine.to(DEV), sd


def test_e2e_scd_forward_backward_vs_oracle_size64():
    """the SCD path (K=3 perception frames, T=5, three decoders, 7-class logits) through
    the same kernels, against the oracle.  A fixed linear functional of the three outputs drives backward."""
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.utils import hot_path_named_params
    ref, mine, sd = _build_pair_k(64, 3, 7)
    ref64 = om.Trainer(om.make_args(num_perception_frame=3, size=64, dataset="SECOND", num_class=7))
    ref64.load_state_dict(sd)
    ref64 = ref64.double()
    pre, post, _ = synth.synth_batch(2, 64, seed=5)
    ref.train(); mine.train(); ref64.train()
    outs_r = ref.update_scd(pre, post)
    outs_64 = ref64.update_scd(pre.double(), post.double())
    outs_d = mine.update_scd(pre.to(DEV), post.to(DEV))
    probes = [synth.synth_tensor(tuple(o.shape), 40 + i, 1.0 / o[0].numel()) for i, o in enumerate(outs_r)]
    sum((o * p).sum() for o, p in zip(outs_r, probes)).backward()
    sum((o * p.double()).sum() for o, p in zip(outs_64, probes)).backward()
    sum((o * p.to(DEV)).sum() for o, p in zip(outs_d, probes)).backward()
    torch.cuda.synchronize()
    for name, o_d, o_r, o_64 in zip(("pre", "post", "change"), outs_d, outs_r, outs_64):
        assert o_d.shape == o_r.shape
        scale = max(1.0, o_64.detach().abs().max().item())
        e_hip = (o_d.detach().cpu().double() - o_64.detach()).abs().max().item() / scale
        e_ref = (o_r.detach().double() - o_64.detach()).abs().max().item() / scale
        print(f"SCD {name}: max rel err vs fp64: hip {e_hip:.3e}  fp32 reference {e_ref:.3e}")
        assert e_hip <= K_NOISE * e_ref + 1e-6, (name, e_hip, e_ref)
    # argmax masks (reference scripts/train_SCD.py:241-246): identical wherever the fp64 top-2 margin is safe
    for o_d, o_r, o_64 in zip(outs_d[:2], outs_r[:2], outs_64[:2]):
        top2 = o_64.detach().topk(2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
        tol = K_NOISE * (o_r.detach().double() - o_64.detach()).abs().max().item() * 2 + 1e-6
        safe = margin > tol
        assert ((o_d.detach().cpu().argmax(1) != o_64.detach().argmax(1)) & safe).sum().item() == 0
    names = [n for n, _ in hot_path_named_params(mine)]
    g_hip = {n: p.grad for n, p in hot_path_named_params(mine)}
    g32 = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    assert set(names) == set(g32.keys())
    _grad_check(names, g_hip, g32, g64)


@pytest.mark.parametrize("gsize", [64, 256])
def test_e2e_scd_vs_reference_golden(gsize, golden_dir):
    """SCD training step (update_scd + the loss of scripts/train_SCD.py:226-229) against the fixtures produced
    by the REAL reference (tests/golden/scd_s{64,256}_b2.npz, oracle/gen_golden.py::run_scd; 256 = the benchmarked
    resolution)."""
    _need_gpu()
    from oracle import synth
    from change3d_amd.model.utils import BCEDiceLoss, ChangeSimilarity, CrossEntropyLoss2d, hot_path_named_params
    G = np.load(os.path.join(golden_dir, f"scd_s{gsize}_b2.npz"), allow_pickle=False)
    size, batch = int(G["meta"][0]), int(G["meta"][1])
    _, mine, _ = _build_pair_k(size, 3, 7)
    pre, post, _ = synth.synth_batch(batch, size, seed=int(G["meta"][3]))
    labels = synth.synth_scd_labels(batch, size, seed=int(G["meta"][3])).to(DEV)
    mine.train()
    pm, qm, cm = mine.update_scd(pre.to(DEV), post.to(DEV))
    lc = labels[:, 2].long()
    pl, ql = labels[:, 0].long() * lc, labels[:, 1].long() * lc
    seg, sim = CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity()
    loss = (seg(pm, pl) + seg(qm, ql)) * 0.5 + BCEDiceLoss(cm, lc.unsqueeze(1).float()) + sim(pm[:, 1:], qm[:, 1:], lc.unsqueeze(1))
    loss.backward()
    torch.cuda.synchronize()
    stride = max(size // 32, 1)
    for k, o in zip(("pre", "post", "change"), (pm, qm, cm)):
        lat = o.detach()[:, :, ::stride, ::stride].cpu().double().numpy()
        scale = max(1.0, np.abs(G[f"{k}_lattice_f64"]).max())
        e_hip = np.abs(lat - G[f"{k}_lattice_f64"]).max() / scale
        e_ref = np.abs(G[f"{k}_lattice"].astype(np.float64) - G[f"{k}_lattice_f64"]).max() / scale
        print(f"SCD {k}: max rel err vs fp64 on lattice: hip {e_hip:.3e}  fp32 reference {e_ref:.3e}")
        assert e_hip <= K_NOISE * e_ref + 1e-6, (k, e_hip, e_ref)
    l_ref, l_64 = float(G["loss"]), float(G["loss_f64"])
    print(f"SCD loss hip {loss.item():.6f} ref32 {l_ref:.6f} ref64 {l_64:.6f}")
    assert abs(loss.item() - l_64) <= K_NOISE * abs(l_ref - l_64) + 1e-4
    named = dict(hot_path_named_params(mine))
    names = [str(n) for n in G["grad_names"]]
    assert set(names) == set(named.keys())
    gn = np.array([named[n].grad.double().norm().item() for n in names])
    eh = np.abs(gn - G["grad_norms_f64"]) / (G["grad_norms_f64"] + 1e-30)
    er = np.abs(G["grad_norms"] - G["grad_norms_f64"]) / (G["grad_norms_f64"] + 1e-30)
    _noise_check(G["grad_names"], eh, er, "SCD grad-norm rel err")


def test_step_is_reproducible():
    """Two forward+backward passes from identical state: activations (hence the loss) must be
    bit-identical and every gradient reproducible up to f32 leaf-gradient atomics.  (Statistics are
    reduced in fixed order or in fixed point; an order-dependent f32 reduction in ONE BatchNorm is
    amplified by the 55 train-mode BN layers behind it into percent-level gradient noise.)"""
    _need_gpu()
    from change3d_amd.model.utils import BCEDiceLoss
    from oracle import synth
    _, mine, _ = _build_pair(64)
    pre, post, tgt = synth.synth_batch(2, 64, seed=3)
    pre, post, tgt = pre.to(DEV), post.to(DEV), tgt.to(DEV)
    mine.train()
    state = {k: v.clone() for k, v in mine.state_dict().items()}
    runs = []
    for _ in range(3):
        mine.load_state_dict(state)
        for p_ in mine.parameters():
            p_.grad = None
        prob = mine.update_bcd(pre, post)
        loss = BCEDiceLoss(prob, tgt)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((prob.detach().clone(), loss.item(),
                     {n: p_.grad.detach().clone() for n, p_ in mine.named_parameters() if p_.grad is not None}))
    p0, l0, g0 = runs[0]
    for p1, l1, g1 in runs[1:]:
        assert torch.equal(p0, p1) and l0 == l1
        worst = max(((g1[n] - g0[n]).double().norm() / (g0[n].double().norm() + 1e-30)).item() for n in g0)
        assert worst < 2e-5, worst


def test_e2e_bf16_tracks_f32():
    """Throughput path (bf16 activations): not bit-parity -- report and bound the drift.  The bound is asserted on the
    well-conditioned (trained-network-like, branch_gain = 0.1) weights, where bf16 storage leaves the change mask within
    IoU 0.95 of the fp32 oracle's (measured 0.96-0.98 on MI355X); on the chaotic default synthetic weights -- one f32 rounding
    of the input moves a typical gradient by 0.7 %, section 1 of DESIGN.md -- the same comparison gives IoU ~0.87 (the
    survey's own bf16-vs-f32 CPU probe: 0.89): reported, with only a sanity floor."""
    _need_gpu()
    from oracle import model as om, synth
    from change3d_amd.model.utils import BCEDiceLoss
    for gain, floor in ((0.1, 0.95), (None, 0.8)):
        ref, mine, _ = _build_pair(64, act_dtype=torch.bfloat16, branch_gain=gain)
        pre, post, tgt = synth.synth_batch(2, 64, seed=0)
        ref.train(); mine.train()
        pr = ref.update_bcd(pre, post)
        lref = om.bce_dice_loss(pr, tgt)
        pd = mine.update_bcd(pre.to(DEV), post.to(DEV))
        ld = BCEDiceLoss(pd, tgt.to(DEV))
        ld.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(pd).all()
        inter = ((pd.cpu() > 0.5) & (pr > 0.5)).sum().item()
        union = ((pd.cpu() > 0.5) | (pr > 0.5)).sum().item()
        iou = inter / max(union, 1)
        print(f"bf16 vs f32 oracle ({'conditioned' if gain else 'default'} weights): max|dp|={(pd.cpu() - pr).abs().max().item():.3e} "
              f"mask IoU={iou:.4f} loss {ld.item():.4f} vs {lref.item():.4f}")
        assert iou > floor, (gain, iou)
        assert abs(ld.item() - lref.item()) < 0.1 * abs(lref.item())
        for n, p in mine.named_parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("flag", ["STAGE_SEPARATE_FINALIZE", "STAGE_NO_WEIGHT_IMAGES", "STAGE_SEPARATE_RESIDUAL", "STAGE_SEPARATE_WGRAD"])
def test_folded_batchnorm_launches_are_bit_identical_end_to_end(flag):
    """Default launch sequence (BatchNorm finalisation / backward coefficients rebuilt by their consumers, csrc/bn_fin.h;
    pointwise weights read as packed LDS images, c3d_pw_pack_weights; residual add in the next block's conv_a) vs the
    unfused sequences of c3d_stage_desc.flags (246 separate finalize launches per step / every GEMM workgroup converts
    the f32 weights itself / c3d_block_out_fwd launches): the bf16 train step produces the same loss, gradients, running
    statistics and num_batches_tracked to the last bit."""
    _need_gpu()
    import contextlib
    import io
    from change3d_amd import ops, synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss
    from change3d_amd.model.x3d import X3DResStage
    outs = []
    # (the workgroup-cooperative kernels -- conv_a / conv_c forward, conv_a data + weight gradient -- take only the default
    # sequence's argument forms and group f32 partial sums differently from the first kernel -- 1e-6, tests/test_ops_gpu.py,
    # tests/test_pw_wg_gpu.py; both runs of this comparison use the first kernel)
    ops.set_option(ops.OPT_PW_CFWD, 0)
    ops.set_option(ops.OPT_PW_CDG, 0)
    try:
        for flags in (0, getattr(ops, flag)):
            args = synth.make_args(size=64, act_dtype=torch.bfloat16)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                net = Trainer(args)
            net.load_state_dict(synth.synth_state_dict(net, seed=5, mask_margin=0.25, branch_gain=0.1))
            net = net.to(DEV).train()
            for m in net.modules():
                if isinstance(m, X3DResStage):
                    m.driver_flags = flags
            pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(3, 64, seed=2))
            loss = BCEDiceLoss(net.update_bcd(pre, post), tgt)
            loss.backward()
            torch.cuda.synchronize()
            outs.append({"loss": loss.detach().cpu(), "grads": {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None},
                         "bufs": {n: b.cpu() for n, b in net.named_buffers()}})
    finally:
        ops.set_option(ops.OPT_PW_CFWD, 3)
        ops.set_option(ops.OPT_PW_CDG, 3)
    a, b = outs
    assert torch.equal(a["loss"], b["loss"]) and torch.isfinite(a["loss"])
    assert a["grads"].keys() == b["grads"].keys() and len(a["grads"]) > 400
    bad = [(n, (a["grads"][n] - b["grads"][n]).abs().max().item()) for n in a["grads"] if not torch.equal(a["grads"][n], b["grads"][n])]
    bad += [(n, -1.0) for n in a["bufs"] if not torch.equal(a["bufs"][n], b["bufs"][n])]
    # weight gradients that end in f32 atomics (stem, depthwise conv_b, SE, perception frames, decoder heads; since round 4
    # conv_a / conv_c of res2 and res3, whose weight gradient is accumulated with LDS atomics inside the data-gradient launch)
    # differ in the last bits between ANY two runs; everything the folded launches compute -- BatchNorm parameter gradients,
    # the res4 pointwise weight gradients that read the coefficients, shortcut convolutions, running statistics -- must be identical
    fused_dw = lambda n: n.endswith(("conv_a.weight", "conv_c.weight")) and (".blocks.1." in n or ".blocks.2." in n)  # noqa: E731
    exact = lambda n: ((".norm" in n or "branch1_norm" in n or n.endswith(("conv_a.weight", "conv_c.weight", "branch1_conv.weight")))  # noqa: E731
                       and ".norm_b.1." not in n and not fused_dw(n))
    if flag == "STAGE_SEPARATE_WGRAD":
        # not a re-association of the same kernels but another kernel variant for conv_c's data gradient (fewer tiles in
        # flight per wave -> another grouping of the f32 partial sums behind the Swish/SE-backward statistics): everything
        # downstream of it last-bit changes of the BatchNorm_b-backward coefficients flip the bf16 rounding of stored gradient
        # elements and the network amplifies that: res4 / decoder tensors (upstream in the backward pass) stay bit-identical,
        # res3 / res2 / stem tensors agree in the L2 sense -- the cancellation-prone BatchNorm weight gradients least
        # (measured with tools/wg_compare.py: worst 2e-2 on a norm_a.weight, typical 1e-4; two runs of ONE variant: 2e-7)
        assert torch.equal(a["loss"], b["loss"])
        rels = {}
        for n in a["grads"]:
            d = (a["grads"][n] - b["grads"][n]).double()
            rels[n] = (d.norm() / a["grads"][n].double().norm().clamp_min(1e-30)).item()
            if ".blocks.3." in n or n.startswith("decoder"):
                assert rels[n] < 1e-5, (n, rels[n])          # (f32 atomics of the leaf gradients only)
        vals = sorted(rels.values())
        assert vals[-1] < 8e-2 and vals[len(vals) * 9 // 10] < 5e-3, (vals[-1], vals[len(vals) * 9 // 10])
        assert all(torch.equal(a["bufs"][n], b["bufs"][n]) for n in a["bufs"])
        return
    wrong = [(n, e) for n, e in bad if e < 0 or exact(n)]
    assert not wrong, (len(wrong), wrong[:8])
    # the fused weight gradients: same bf16 operands, f32 sums in another order (and vs the separate c3d_pw_wgrad kernel)
    for n in a["grads"]:
        if fused_dw(n):
            rel = ((a["grads"][n] - b["grads"][n]).abs().max() / a["grads"][n].abs().max().clamp_min(1e-30)).item()
            assert rel < 2e-5, (n, rel)
    for n, e in bad:
        assert e <= 1e-5 * b["grads"][n].abs().max().item(), (n, e)
    assert sum(1 for n in a["grads"] if exact(n)) > 270


def test_conv_c_forward_on_the_cooperative_kernel_end_to_end():
    """C3D_OPT_PW_CFWD (csrc/pw_cfwd.hip: conv_c forward of the training path on workgroup-cooperative tiles) against the
    wave-private-tile kernel, through a whole bf16 BCD train step on conditioned weights (128 x 128, so that res2 / res3 / res4
    all take the new kernel).  conv_c's OUTPUT is bit-identical between the two (tests/test_ops_gpu.py::
    test_conv_c_forward_kernels_agree_bit_for_bit); its BatchNorm_c statistics group their f32 partial sums differently (1e-7),
    which moves last bits of a few scale / shift values, flips isolated bf16 roundings and is amplified down the network to the
    level of bf16 quantisation noise -- the same signature as C3D_OPT_MASK_IN_DGRAD (test below).  Pinned: the loss to 2e-3, the
    change probabilities to 2e-2 (bf16 activations), every gradient finite, parameter gradients within sqrt(2) x the bf16 gradient
    noise of this network (median < 1e-1, flat-gradient cosine > 0.99), running statistics to 5e-3 of their largest entry."""
    _need_gpu()
    import contextlib
    import io
    from change3d_amd import ops, synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss
    outs = []
    try:
        for opt in (0, 1):
            ops.set_option(ops.OPT_PW_CFWD, opt)
            args = synth.make_args(size=128, act_dtype=torch.bfloat16)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                net = Trainer(args)
            net.load_state_dict(synth.synth_state_dict(net, seed=5, mask_margin=0.25, branch_gain=0.1))
            net = net.to(DEV).train()
            pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(3, 128, seed=2))
            prob = net.update_bcd(pre, post)
            loss = BCEDiceLoss(prob, tgt)
            loss.backward()
            torch.cuda.synchronize()
            outs.append({"loss": loss.detach().cpu(), "prob": prob.detach().float().cpu(),
                         "grads": {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None},
                         "bufs": {n: b.float().cpu() for n, b in net.named_buffers()}})
    finally:
        ops.set_option(ops.OPT_PW_CFWD, 3)
        ops.set_option(ops.OPT_PW_CDG, 3)
    a, b = outs
    assert torch.isfinite(b["loss"]) and abs(a["loss"].item() - b["loss"].item()) < 2e-3 * abs(a["loss"].item()), (a["loss"], b["loss"])
    assert (a["prob"] - b["prob"]).abs().max().item() < 2e-2
    assert a["grads"].keys() == b["grads"].keys() and len(a["grads"]) > 400
    worst_buf = max(((a["bufs"][n] - b["bufs"][n]).abs().max().item() / (a["bufs"][n].abs().max().item() + 1e-3), n) for n in a["bufs"])
    assert worst_buf[0] < 5e-3, worst_buf   # running statistics of layers BELOW the first flipped rounding see bf16-level differences
    rels = []
    for n in a["grads"]:
        assert torch.isfinite(b["grads"][n]).all(), n
        d = (a["grads"][n] - b["grads"][n]).double()
        rels.append((d.norm() / a["grads"][n].double().norm().clamp_min(1e-30)).item())
    rels.sort()
    print(f"  PW_CFWD 0 vs 1: |d loss| {abs(a['loss'].item() - b['loss'].item()):.2e}, parameter-gradient rel-L2 median {rels[len(rels) // 2]:.1e}, "
          f"90 % {rels[len(rels) * 9 // 10]:.1e}, worst {rels[-1]:.1e}")
    # Two bf16 runs whose roundings have decorrelated differ by sqrt(2) x the bf16 gradient noise of this network (median 4e-2
    # against the f64 oracle on conditioned weights): measured 5.7e-2 median, 1.1e-1 at 90 %, 2.3e-1 worst;
    # the flat gradient stays aligned
    assert rels[len(rels) // 2] < 1e-1 and rels[len(rels) * 9 // 10] < 2.5e-1 and rels[-1] < 0.6, (rels[len(rels) // 2], rels[len(rels) * 9 // 10], rels[-1])
    fa = torch.cat([a["grads"][n].double().flatten() for n in sorted(a["grads"])])
    fb = torch.cat([b["grads"][n].double().flatten() for n in sorted(b["grads"])])
    cos = (fa @ fb / (fa.norm() * fb.norm())).item()
    assert cos > 0.99, cos


def test_cooperative_data_and_weight_gradients_end_to_end():
    """C3D_OPT_PW_CDG (csrc/pw_cdgrad.hip: conv_a / conv_c data gradients WITH their weight gradients on workgroup-cooperative
    tiles, every stage width) against the wave-private kernels + separate weight-gradient launches, through a whole bf16 BCD train
    step on conditioned weights (128 x 128: res2 / res3 / res4 all take the new kernels).  The forward pass is the same code:
    loss and probabilities equal to the last bit.  The data gradients are bit-identical per launch (tests/test_pw_wg_gpu.py); the
    BatchNorm-backward sums group their f32 partial sums differently, which flips isolated bf16 roundings of stored gradient
    rows further down: parameter gradients agree to a median of 2e-3 (bound 1e-2; worst tensor 2e-2, bound 8e-2), flat-gradient
    cosine > 0.9999; the weight gradients of the last res4 block (the first the backward pass reaches) agree to 1e-3."""
    _need_gpu()
    import contextlib
    import io
    from change3d_amd import ops, synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss
    outs = []
    try:
        for opt in (0, 3):
            ops.set_option(ops.OPT_PW_CDG, opt)
            args = synth.make_args(size=128, act_dtype=torch.bfloat16)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                net = Trainer(args)
            net.load_state_dict(synth.synth_state_dict(net, seed=5, mask_margin=0.25, branch_gain=0.1))
            net = net.to(DEV).train()
            pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(3, 128, seed=2))
            prob = net.update_bcd(pre, post)
            loss = BCEDiceLoss(prob, tgt)
            loss.backward()
            torch.cuda.synchronize()
            outs.append({"loss": loss.detach().cpu(), "prob": prob.detach().float().cpu(),
                         "grads": {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}})
    finally:
        ops.set_option(ops.OPT_PW_CDG, 3)
    a, b = outs
    assert torch.equal(a["loss"], b["loss"]) and torch.equal(a["prob"], b["prob"]), "the forward pass must not change"
    assert a["grads"].keys() == b["grads"].keys() and len(a["grads"]) > 400
    rels = {}
    for n in a["grads"]:
        assert torch.isfinite(b["grads"][n]).all(), n
        d = (a["grads"][n] - b["grads"][n]).double()
        rels[n] = (d.norm() / a["grads"][n].double().norm().clamp_min(1e-30)).item()
    rs = sorted(rels.values())
    print(f"  PW_CDG 0 vs 3: parameter-gradient rel-L2 median {rs[len(rs) // 2]:.1e}, 90 % {rs[len(rs) * 9 // 10]:.1e}, worst {rs[-1]:.1e}")
    # measured: median 2.1e-3, 90 % 4.7e-3, worst 1.8e-2 (only the backward sums differ: far less decorrelation than a changed forward)
    assert rs[len(rs) // 2] < 1e-2 and rs[len(rs) * 9 // 10] < 2.5e-2 and rs[-1] < 8e-2, (rs[len(rs) // 2], rs[len(rs) * 9 // 10], rs[-1])
    fa = torch.cat([a["grads"][n].double().flatten() for n in sorted(a["grads"])])
    fb = torch.cat([b["grads"][n].double().flatten() for n in sorted(b["grads"])])
    cos = (fa @ fb / (fa.norm() * fb.norm())).item()
    assert cos > 0.9999, cos
    # the last res4 block is the first one the backward pass reaches: its gradients saw no flipped rounding yet
    last = [n for n in rels if ".blocks.3." in n and ".res_blocks.24." in n and ("conv_a.weight" in n or "conv_c.weight" in n)]
    assert last, [n for n in rels if ".blocks.3." in n][:5]
    for n in last:
        assert rels[n] < 1e-3, (n, rels[n])


def test_block_output_backward_folded_into_conv_a_agrees_with_the_separate_launches_end_to_end():
    """C3D_OPT_MASK_IN_DGRAD = 3 (default: c3d_block_out_bwd of a block runs in the epilogue of the conv_a data gradient of the
    block above it -- mask and BatchNorm_c-backward sums, c3d_pw_args.add_sums / C3D_WG_MASKSUM) against = 1 (mask only where the
    weight gradient is fused, the sums by c3d_block_out_bwd launches): same forward, so the same loss to the last bit; the
    BatchNorm_c-backward sums are accumulated in another order (per lane and wave in f32, then f64), which moves last bits of the
    coefficients, flips bf16 roundings of stored gradient rows and is amplified down the 24 blocks below to the level of bf16
    quantisation noise (median 5e-3, measured): the test pins that it ENTERS as a last-bit difference and stays at that level."""
    _need_gpu()
    import contextlib
    import io
    from change3d_amd import ops, synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss
    outs = []
    try:
        for opt in (1, 3):
            ops.set_option(ops.OPT_MASK_IN_DGRAD, opt)
            args = synth.make_args(size=64, act_dtype=torch.bfloat16)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                net = Trainer(args)
            net.load_state_dict(synth.synth_state_dict(net, seed=5, mask_margin=0.25, branch_gain=0.1))
            net = net.to(DEV).train()
            pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(3, 64, seed=2))
            loss = BCEDiceLoss(net.update_bcd(pre, post), tgt)
            loss.backward()
            torch.cuda.synchronize()
            outs.append({"loss": loss.detach().cpu(), "grads": {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None},
                         "bufs": {n: b.cpu() for n, b in net.named_buffers()}})
    finally:
        ops.set_option(ops.OPT_MASK_IN_DGRAD, 3)
    a, b = outs
    assert torch.equal(a["loss"], b["loss"]) and torch.isfinite(a["loss"])
    assert a["grads"].keys() == b["grads"].keys() and len(a["grads"]) > 400
    assert all(torch.equal(a["bufs"][n], b["bufs"][n]) for n in a["bufs"])
    rels = {}
    for n in a["grads"]:
        assert torch.isfinite(b["grads"][n]).all(), n
        d = (a["grads"][n] - b["grads"][n]).double()
        rels[n] = (d.norm() / a["grads"][n].double().norm().clamp_min(1e-30)).item()
        if n.startswith("decoder"):
            assert rels[n] < 1e-5, (n, rels[n])          # upstream of every residual stage in the backward pass
    import re as _re
    by = {}
    for n, r in rels.items():
        m = _re.search(r"blocks\.(\d+)\.res_blocks\.(\d+)", n)
        by.setdefault((int(m.group(1)), int(m.group(2))) if m else (-1, 0), []).append(r)
    med = {k: sorted(v)[len(v) // 2] for k, v in by.items()}
    print("  per block of res4, last to first (median rel-L2):", " ".join(f"{med[k]:.1e}" for k in sorted(by, reverse=True) if k[0] == 3))
    # where the difference ENTERS it is a last-bit one: the last block keeps its own c3d_block_out_bwd in both modes, the
    # next two differ by f32 summation order only; from there on bf16 storage amplifies it to its quantisation level (~0.5 %)
    last = max(k[1] for k in by if k[0] == 3)
    assert max(by[(3, last)]) < 1e-6 and max(by[(3, last - 1)]) < 1e-6, (by[(3, last)], by[(3, last - 1)])
    assert med[(3, last - 2)] < 1e-4, med[(3, last - 2)]
    vals = sorted(rels.values())
    print(f"fold vs separate: worst {vals[-1]:.2e}, 90th percentile {vals[len(vals) * 9 // 10]:.2e}, median {vals[len(vals) // 2]:.2e}")
    assert vals[-1] < 8e-2 and vals[len(vals) * 9 // 10] < 2e-2 and vals[len(vals) // 2] < 1e-2, (vals[-1], vals[len(vals) * 9 // 10], vals[len(vals) // 2])
