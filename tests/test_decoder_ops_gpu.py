"""Op-level parity of the ChangeDecoder's ConvTranspose2d k4 s2 p1 kernels (reference model/change_decoder.py:30-45),
called through the C ABI, against torch-CPU fp32 `conv_transpose2d` and its autograd: forward (+bias +skip frame),
data gradient, weight gradient.  bf16 storage runs the MFMA kernels (csrc/convt_mfma.hip: weights are rounded to
bf16 there, so the reference uses bf16-rounded weights too); f32 storage runs the scalar kernels (csrc/decoder.hip)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import DEV, DTYPES, close, q, rnd

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,B,h,w", [(24, 2, 8, 16), (24, 3, 13, 37), (48, 2, 9, 20), (24, 1, 4, 70)])
def test_convT4s2_fwd_bwd_data_wgrad(dtype, C, B, h, w):
    _need_gpu()
    from change3d_amd import ops
    dt = ops.dt_code(dtype)
    bf = dtype == torch.bfloat16
    x = q(rnd((B, h, w, C), 1), dtype)                       # channels-last layer input
    wt = rnd((C, C, 4, 4), 2, 0.2)
    wq = wt.to(torch.bfloat16).float() if bf else wt          # the MFMA kernels round the weights to bf16
    bias = rnd((C,), 3, 0.5)
    T = 3
    skip_full = q(rnd((B, T, 2 * h, 2 * w, C), 4), dtype)     # NDHWC tensor; frame 1 is the skip connection
    dout = q(rnd((B, 2 * h, 2 * w, C), 5), dtype)
    # ---- reference
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, bias, stride=2, padding=1) + skip_full[:, 1].permute(0, 3, 1, 2)
    yr.backward(dout.permute(0, 3, 1, 2))
    # ---- device
    xd, sd, dd = x.to(DEV).to(dtype), skip_full.to(DEV).to(dtype), dout.to(DEV).to(dtype)
    wd_, bd = wt.to(DEV), bias.to(DEV)
    out = torch.empty((B, 2 * h, 2 * w, C), dtype=dtype, device=DEV)
    frame = sd[:, 1]
    ops.convT_fwd(xd, wd_, bd, frame.data_ptr(), sd.stride(0), out, B, h, w, C, dt)
    din = torch.empty((B, h, w, C), dtype=dtype, device=DEV)
    ops.convT_bwd_data(dd, wd_, din, B, h, w, C, dt)
    gw = torch.zeros((C, C, 4, 4), dtype=torch.float32, device=DEV)
    if bf:
        ops.convT_wgrad(xd, dd, gw, B, h, w, C, dt)
    else:
        ops.pw_wgrad(xd, dd, gw, M=B * h * w, K=C, N=C, dw_sn=C * 16, dw_sk=16, dtype=dt, row_mode=ops.ROWS_S2SHIFT,
                     H=2 * h, W=2 * w, taps=16, dw_tap_stride=1)
    torch.cuda.synchronize()
    sc = float(yr.abs().max())
    close(out.permute(0, 3, 1, 2), yr, dtype, "convT forward", scale=sc)
    close(din.permute(0, 3, 1, 2), xr.grad, dtype, "convT data gradient", scale=float(xr.grad.abs().max()))
    rel = (gw.cpu() - wr.grad).norm().item() / wr.grad.norm().item()
    assert rel < (2e-5 if not bf else 3e-3), ("convT weight gradient rel-L2", rel)
    # second accumulation into the same buffer (+=)
    if bf:
        ops.convT_wgrad(xd, dd, gw, B, h, w, C, dt)
        torch.cuda.synchronize()
        assert (gw.cpu() - 2 * wr.grad).norm().item() / wr.grad.norm().item() < 6e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("NC,sig", [(1, True), (7, False), (2, True)])
@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (3, 13, 37), (1, 40, 100)])
def test_head3x3_fwd_bwd(dtype, NC, sig, B, H, W):
    """Final 3x3 convolution 24 -> NC (+ sigmoid for the change head; reference model/change_decoder.py:46-55,78-81) against
    torch-CPU `conv2d` and its autograd.  bf16 storage runs the matrix-core kernels (csrc/head_mfma.hip: weights and the
    logit gradient are rounded to bf16 there), f32 storage the scalar kernels (csrc/decoder.hip); ragged tiles included."""
    _need_gpu()
    from change3d_amd import ops
    dt = ops.dt_code(dtype)
    bf = dtype == torch.bfloat16
    C = 24
    x = q(rnd((B, H, W, C), 11), dtype)
    wt = rnd((NC, C, 3, 3), 12, 0.2)
    wq = wt.to(torch.bfloat16).float() if bf else wt
    dout = rnd((B, NC, H, W), 13)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    logit = F.conv2d(xr, wr, padding=1)
    yr = torch.sigmoid(logit) if sig else logit
    yr.backward(dout)
    xd, wd_, dd = x.to(DEV).to(dtype), wt.to(DEV), dout.to(DEV)
    out = torch.full((B, NC, H, W), float("nan"), device=DEV)
    ops.head_fwd(xd, wd_, out, B, H, W, C, NC, sig, dt)
    dx = torch.full((B, H, W, C), float("nan"), dtype=dtype, device=DEV)
    dw = torch.ones((NC, C, 3, 3), device=DEV)                      # accumulate semantics
    ops.head_bwd(dd, out if sig else None, xd, wd_, dx, dw, B, H, W, C, NC, sig, dt)
    torch.cuda.synchronize()
    tol_out = 2e-5 if not bf else 2e-3
    assert (out.cpu() - yr.detach()).abs().max().item() < tol_out * max(1.0, float(yr.detach().abs().max()))
    close(dx.permute(0, 3, 1, 2), xr.grad, dtype, "head data gradient", scale=float(xr.grad.abs().max()))
    rel = ((dw.cpu() - 1.0) - wr.grad).norm().item() / wr.grad.norm().item()
    assert rel < (2e-5 if not bf else 4e-3), ("head weight gradient rel-L2", rel)
    if bf:   # the scalar kernels of the same storage type (C3D_OPT_CONVT_MFMA = 0) agree with the matrix-core ones
        ops.set_option(ops.OPT_CONVT_MFMA, 0)
        try:
            out2 = torch.empty_like(out)
            ops.head_fwd(xd, wd_, out2, B, H, W, C, NC, sig, dt)
            dx2 = torch.empty_like(dx)
            dw2 = torch.zeros_like(dw)
            ops.head_bwd(dd, out2 if sig else None, xd, wd_, dx2, dw2, B, H, W, C, NC, sig, dt)
            torch.cuda.synchronize()
        finally:
            ops.set_option(ops.OPT_CONVT_MFMA, 1)
        assert (out - out2).abs().max().item() < 1.5e-2 * max(1.0, float(out2.abs().max()))   # (f32 vs bf16-rounded weights)
        assert ((dx.float() - dx2.float()).norm() / dx2.float().norm()).item() < 1e-2
        assert (((dw - 1.0) - dw2).norm() / dw2.norm()).item() < 1e-2
