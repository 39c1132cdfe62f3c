"""Step-shell pieces on the GPU (SURVEY.md 8(f).3 and the rows the round-1 review found untested): the frame tap
(`c3d_frame_scatter`), checkpoint round trip between the HIP mirror and the oracle, `update_bda` (T=4)."""
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
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_frame_scatter_op(dtype, accumulate):
    _need_gpu()
    from change3d_amd import ops, synthetic as synth
    B, T, H, W, C = 3, 5, 9, 7, 48
    src = synth.synth_tensor((B, H, W, C), 1).to(DEV).to(dtype)
    dst = synth.synth_tensor((B, T, H, W, C), 2).to(DEV).to(dtype)
    want = dst.clone()
    want[:, 3] = (want[:, 3].float() + src.float()).to(dtype) if accumulate else src
    ops.frame_scatter(src, dst, B, T, H * W, C, 3, accumulate, ops.dt_code(dtype))
    torch.cuda.synchronize()
    assert torch.equal(dst, want)


@pytest.mark.parametrize("last", [False, True])
def test_tap_frames_backward_matches_plain_indexing(last):
    """`tap_frames` against the reference's plain `x[:, :, k]` (model/trainer.py:136-139) under autograd;
    `last`: x has no other consumer (the res4 case: the gradient buffer is created inside the tap)."""
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import tap_frames
    B, C, T, H, W = 2, 24, 5, 8, 8
    base = synth.synth_tensor((B, T, H, W, C), 3).to(DEV).permute(0, 4, 1, 2, 3)   # channels-last storage
    gy = synth.synth_tensor((B, T, H, W, C), 4).to(DEV).permute(0, 4, 1, 2, 3)
    gfs = [synth.synth_tensor((B, H, W, C), 5 + i).to(DEV).permute(0, 3, 1, 2) for i in range(3)]
    xr = base.clone().requires_grad_(True)
    fr = [xr[:, :, 1 + i] for i in range(3)]
    tot = sum((f * g).sum() for f, g in zip(fr, gfs)) + (0 if last else (xr * 2.0 * gy).sum())
    tot.backward()
    xd = base.clone().requires_grad_(True)
    y, fd = tap_frames(xd * 1.0, 1, 3)
    tot = sum((f * g).sum() for f, g in zip(fd, gfs)) + (0 if last else (y * 2.0 * gy).sum())
    tot.backward()
    assert torch.allclose(xd.grad, xr.grad, rtol=1e-6, atol=1e-6)


def test_tap_frames_copy_mode_leaves_the_incoming_gradient_untouched():
    """`trainer.TAP_INPLACE = False` (for hooks / retain_grad on stage outputs): the gradient that arrives for x is
    not written -- a reference kept by a tensor hook still holds the next stage's gradient alone -- and x.grad is the
    same as in the default in-place mode."""
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model import trainer as tr
    B, C, T, H, W = 2, 24, 5, 8, 8
    base = synth.synth_tensor((B, T, H, W, C), 3).to(DEV).permute(0, 4, 1, 2, 3)
    gy = synth.synth_tensor((B, T, H, W, C), 4).to(DEV).permute(0, 4, 1, 2, 3)
    gfs = [synth.synth_tensor((B, H, W, C), 5 + i).to(DEV).permute(0, 3, 1, 2) for i in range(3)]
    grads, seen = [], []
    try:
        for inplace in (True, False):
            tr.TAP_INPLACE = inplace
            xd = base.clone().requires_grad_(True)
            y, fd = tr.tap_frames(xd * 1.0, 1, 3)
            kept = []
            y.register_hook(lambda g, kept=kept: kept.append((g, g.clone())))
            (sum((f * g).sum() for f, g in zip(fd, gfs)) + (y * 2.0 * gy).sum()).backward()
            grads.append(xd.grad.clone())
            seen.append(torch.equal(kept[0][0], kept[0][1]))
    finally:
        tr.TAP_INPLACE = True
    assert torch.equal(grads[0], grads[1])
    assert seen[1], "copy mode wrote into the gradient it received"


def test_checkpoint_round_trip_mirror_oracle_mirror(tmp_path):
    """reference scripts/train_BCD.py:333-349 (checkpoint layout) / model/utils.py:205-232 (resume): a checkpoint
    written from the HIP mirror after a train step strict-loads into the oracle (= the reference module tree) and
    the oracle's eval output agrees; loaded back through the mirror's `load_checkpoint` into a fresh HIP model
    WITH a parameter arena the eval output is bit-identical."""
    _need_gpu()
    from types import SimpleNamespace
    from oracle import model as om
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import (BCEDiceLoss, FusedAdam, ParamArena, hot_path_named_params,
                                          load_checkpoint)
    S, B = 64, 2
    args = synth.make_args(size=S)
    net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
    net = net.to(DEV).train()
    arena = ParamArena(hot_path_named_params(net), torch.device(DEV))
    opt = FusedAdam(arena, lr=2e-4)
    pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(B, S, seed=0))
    for _ in range(2):
        opt.zero_grad()
        loss = BCEDiceLoss(net.update_bcd(pre, post), tgt)
        loss.backward()
        opt.step()
    net.eval()
    with torch.no_grad():
        p_hip = net.update_bcd(pre, post).clone()
    ckpt = {"epoch": 3, "arch": str(net), "state_dict": net.state_dict(), "optimizer": opt.state_dict(),
            "loss_train": float(loss), "loss_val": 0.0, "F_train": 0.0, "F_val": 0.0, "lr": 2e-4}
    torch.save(ckpt, os.path.join(tmp_path, "checkpoint.pth.tar"))
    # ---- into the oracle (reference module tree), strict
    state = torch.load(os.path.join(tmp_path, "checkpoint.pth.tar"), map_location="cpu")
    ora = om.Trainer(args)
    ora.load_state_dict(state["state_dict"], strict=True)
    ora.eval()
    with torch.no_grad():
        p_ora = ora.update_bcd(pre.cpu(), post.cpu())
    ora64 = om.Trainer(args)
    ora64.load_state_dict(state["state_dict"], strict=True)
    ora64 = ora64.double().eval()
    with torch.no_grad():
        p_64 = ora64.update_bcd(pre.cpu().double(), post.cpu().double())
    e_hip, e_ref = (p_hip.cpu().double() - p_64).abs().max().item(), (p_ora.double() - p_64).abs().max().item()
    print(f"checkpoint -> oracle: eval max|p - p_fp64| hip {e_hip:.3e}  fp32 oracle {e_ref:.3e}")
    assert e_hip <= 4.0 * e_ref + 1e-5, (e_hip, e_ref)      # same K_NOISE rule as tests/test_model_gpu.py
    assert state["optimizer"]["step"] == 2 and state["epoch"] == 3
    # ---- oracle-written checkpoint back into a fresh mirror through load_checkpoint (resume path)
    torch.save({"epoch": 3, "state_dict": ora.state_dict()}, os.path.join(tmp_path, "checkpoint.pth.tar"))
    net2 = Trainer(args).to(DEV)
    arena2 = ParamArena(hot_path_named_params(net2), torch.device(DEV))
    start_epoch, cur_iter = load_checkpoint(SimpleNamespace(resume=True), net2, str(tmp_path), 100)
    assert (start_epoch, cur_iter) == (3, 300)
    arena2.check_grads_attached()
    for p, o in zip(arena2.params, arena2.offsets):   # parameters still live in the arena after the load
        assert p.data_ptr() == arena2.flat_param.data_ptr() + 4 * o
    net2.eval()
    with torch.no_grad():
        p_back = net2.update_bcd(pre, post)
    assert torch.equal(p_back, p_hip)
    assert load_checkpoint(SimpleNamespace(resume=None), net2, str(tmp_path), 100) == (0, 0)


def test_e2e_bda_forward_backward_vs_oracle_size64():
    """`Trainer.update_bda` (reference model/trainer.py:268-290): K=2 perception frames, T=4, two decoders."""
    _need_gpu()
    from oracle import model as om
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import hot_path_named_params
    from test_model_gpu import K_NOISE, _grad_check
    S, B = 64, 2
    mk = lambda: om.make_args(num_perception_frame=2, size=S, dataset="xBD", num_class=5)  # noqa: E731
    ref = om.Trainer(mk())
    sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25)
    ref.load_state_dict(sd)
    ref64 = om.Trainer(mk())
    ref64.load_state_dict(sd)
    ref64 = ref64.double()
    mine = Trainer(mk())
    mine.load_state_dict(sd)
    mine = mine.to(DEV)
    pre, post, _ = synth.synth_batch(B, S, seed=7)
    ref.train(); mine.train(); ref64.train()
    outs_r, outs_64 = ref.update_bda(pre, post), ref64.update_bda(pre.double(), post.double())
    outs_d = mine.update_bda(pre.to(DEV), post.to(DEV))
    probes = [synth.synth_tensor(tuple(o.shape), 60 + i, 1.0 / o[0].numel()) for i, o in enumerate(outs_r)]
    sum((o * p).sum() for o, p in zip(outs_r, probes)).backward()
    sum((o * p.double()).sum() for o, p in zip(outs_64, probes)).backward()
    sum((o * p.to(DEV)).sum() for o, p in zip(outs_d, probes)).backward()
    torch.cuda.synchronize()
    for name, o_d, o_r, o_64 in zip(("cls", "loc"), outs_d, outs_r, outs_64):
        assert o_d.shape == o_r.shape
        scale = max(1.0, o_64.detach().abs().max().item())
        e_hip = (o_d.detach().cpu().double() - o_64.detach()).abs().max().item() / scale
        e_ref = (o_r.detach().double() - o_64.detach()).abs().max().item() / scale
        print(f"BDA {name}: max rel err vs fp64: hip {e_hip:.3e}  fp32 reference {e_ref:.3e}")
        assert e_hip <= K_NOISE * e_ref + 1e-6, (name, e_hip, e_ref)
    names = [n for n, _ in hot_path_named_params(mine)]
    g_hip = {n: p.grad for n, p in hot_path_named_params(mine)}
    g32 = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    assert set(names) == set(g32.keys())
    _grad_check(names, g_hip, g32, g64)


@pytest.mark.parametrize("shape", [(5, 33, 40), (4, 64, 64), (3, 17, 23)])
@pytest.mark.parametrize("imagenet", [False, True])
def test_device_input_pipeline_bit_exact_vs_numpy_oracle(shape, imagenet):
    """flip / exchange / normalize / to_tensor on the GPU (c3d_bcd_preprocess) against the numpy
    restatement of reference data/transforms.py:100-154 -- bit-exact (byte shuffling + two IEEE f32 divisions)."""
    _need_gpu()
    from oracle import transforms as ot
    from change3d_amd.data.transforms import BCDTransforms, DeviceBatchTransform, draw_augmentation_flags
    B, H, W = shape
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(B, H, W, 6), dtype=np.uint8)
    lab = (rng.integers(0, 3, size=(B, H, W)) * 127 + (rng.random((B, H, W)) < 0.3)).astype(np.uint8)  # 0,1,127,128,254,255
    flags = draw_augmentation_flags(B, rng)
    flags[0] = (1, 1, 1)
    if B > 1:
        flags[1] = (0, 0, 0)
    mean, std = (BCDTransforms.IMAGENET_MEAN, BCDTransforms.IMAGENET_STD) if imagenet else (BCDTransforms.DEFAULT_MEAN, BCDTransforms.DEFAULT_STD)
    want_pre, want_post, want_lab = ot.bcd_transform_batch(img, lab, flags, mean, std)
    tf = DeviceBatchTransform(DEV, mean, std)
    pre, post, lb = tf(img, lab, flags)
    torch.cuda.synchronize()
    assert np.array_equal(pre.cpu().numpy(), want_pre) and np.array_equal(post.cpu().numpy(), want_post)
    assert np.array_equal(lb.cpu().numpy(), want_lab)
    pre2, post2, none = tf(img)                    # validation transform: no flags, no label
    w2 = ot.bcd_transform_batch(img, lab, np.zeros((B, 3), np.uint8), mean, std)
    assert none is None and np.array_equal(pre2.cpu().numpy(), w2[0]) and np.array_equal(post2.cpu().numpy(), w2[1])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3.0e-1)])
def test_eval_folded_bn_matches_unfolded_eval(dtype, tol):
    """SURVEY.md 8(f).4 (reference scripts/train_BCD.py:92-154 `val()`): the inference path with BatchNorm folded
    into the conv weights (c3d_stage_fold_bn / c3d_stage_fwd_folded) against the same eval forward with the
    BatchNorms applied from their running statistics; the folded copy must follow weight updates."""
    _need_gpu()
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.x3d import X3DResStage
    S, B = 64, 3
    args = synth.make_args(size=S)
    args.act_dtype = dtype
    net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25, branch_gain=0.3))
    net = net.to(DEV).eval()
    stages = [m for m in net.modules() if isinstance(m, X3DResStage)]
    assert all(s.fold_bn_eval for s in stages)
    pre, post, _ = (t.to(DEV) for t in synth.synth_batch(B, S, seed=2))

    def run(fold):
        for s in stages:
            s.fold_bn_eval = fold
        with torch.no_grad():
            return net.update_bcd(pre, post).clone()

    p_fold, p_plain = run(True), run(False)
    assert all(s._fold is not None and s._fold_valid for s in stages[:3])
    err = (p_fold - p_plain).abs().max().item()
    print(f"folded vs unfolded eval ({dtype}): max|dp| {err:.3e} mean|dp| {(p_fold - p_plain).abs().mean().item():.3e}")
    assert err < tol and (p_fold - p_plain).abs().mean().item() < tol / 10 and 0.05 < float(p_plain.std())
    # weights change (load_state_dict): the folded copy must be rebuilt
    net.load_state_dict(synth.synth_state_dict(net, seed=17, mask_margin=0.25, branch_gain=0.3))
    p2_fold, p2_plain = run(True), run(False)
    assert (p2_plain - p_plain).abs().max().item() > 0.2      # the new weights give a different answer ...
    assert (p2_fold - p2_plain).abs().mean().item() < tol / 10      # ... and the folded path follows them
    # train() / eval() round trip invalidates too
    net.train(); net.eval()
    assert not any(s._fold_valid for s in stages)


@pytest.mark.parametrize("script,extra", [
    ("change3d_amd.scripts.train_BCD", ["--max_steps", "8", "--batch_size", "2", "--in_height", "64", "--in_width", "64",
                                        "--synthetic_pairs", "8", "--act_dtype", "bf16"]),
    ("change3d_amd.scripts.train_CC", ["--max_steps", "4", "--batch_size", "2", "--in_height", "64", "--in_width", "64",
                                       "--print_freq", "1", "--act_dtype", "f32", "--eval_pairs", "2", "--beam_size", "3"]),
])
def test_training_script_mirrors_run_end_to_end(script, extra, tmp_path):
    """The `scripts/train_*.py`-shaped drivers (reference scripts/train_BCD.py:179-360, scripts/train_CC.py:75-168) run
    a few iterations on the GPU: raw uint8 batch -> device input pipeline -> step -> checkpoint (BCD)."""
    _need_gpu()
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", script] + extra
    if script.endswith("train_BCD"):
        cmd += ["--save_dir", str(tmp_path)]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-2000:]
    text = out.stdout + out.stderr
    assert "nan" not in text.lower()
    if script.endswith("train_BCD"):
        saved = [f for _, _, fs in os.walk(tmp_path) for f in fs]
        assert "checkpoint.pth.tar" in saved and "best_model.pth" in saved, saved
    else:
        assert "Loss:" in text and "evaluate: 2 pairs, beam 3" in text
