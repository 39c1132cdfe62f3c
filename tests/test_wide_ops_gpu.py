"""Op-level parity of the WIDE pointwise GEMM / weight-gradient kernels (csrc/pw_wide.hip: K or N above 224 channels --
the X3D res5 stage, 96/192 -> 432 -> 192, executed only by the change-captioning path, reference model/trainer.py:120-124;
and the caption decoder's linear layers, reference model/caption_decoder.py) through the same C ABI entry points
(c3d_pw_gemm / c3d_pw_wgrad) against plain torch-CPU fp32, every fused prologue / epilogue, both storage types."""
import pytest
import torch

from test_ops_gpu import DEV, DTYPES, close, padc, q, rnd

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K,N,bias", [(192, 432, False), (432, 192, False), (96, 432, False), (192, 576, True), (192, 501, True)])
def test_wide_plain_stats_bias(dtype, K, N, bias):
    _need_gpu()
    from change3d_amd import ops
    M = 64 * 5 + 23
    Kp, Np = ops.cpad(K), ops.cpad(N)
    x = q(rnd((M, K), 1), dtype)
    w = rnd((N, K), 2, 0.1)
    bv = rnd((N,), 3, 0.5) if bias else None
    ref = x @ w.t() + (bv if bias else 0.0)
    y = torch.full((M, Np), float("nan"), dtype=dtype, device=DEV)
    stats = torch.zeros(ops.STAT_STRIPES * 2 * N, dtype=torch.float64, device=DEV)
    ops.pw_gemm(padc(x, Kp).to(DEV, dtype).contiguous(), w.to(DEV), y, M=M, K=K, N=N, w_sn=K, w_sk=1, dtype=ops.dt_code(dtype),
                epi_mode=ops.EPI_STATS, stats=stats, bias=bv.to(DEV) if bias else None)
    torch.cuda.synchronize()
    close(y[:, :N], ref, dtype, "y", scale=ref.abs().max().item())
    if Np > N:
        assert (y[:, N:].float() == 0).all(), "pad channels must be zero"
    yq = y[:, :N].float().cpu().double()
    s = stats.cpu().view(ops.STAT_STRIPES, 2 * N).sum(0)
    assert torch.allclose(s[:N], yq.sum(0), rtol=1e-5, atol=2e-3), "column sums"
    assert torch.allclose(s[N:], (yq * yq).sum(0), rtol=1e-5, atol=2e-3), "column sums of squares"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows", [48, 64, 100])
def test_wide_bn_se_swish_prologue(dtype, rows):
    _need_gpu()
    from change3d_amd import ops
    B, K, N = 3, 432, 192
    M = B * rows
    Kp = ops.cpad(K)
    x = q(rnd((M, K), 3), dtype)
    w = rnd((N, K), 4, 0.08)
    scale, shift = rnd((K,), 5).abs() + 0.5, rnd((K,), 6, 0.3)
    gate = torch.sigmoid(rnd((B, K), 7))
    v = (x * scale + shift).view(B, rows, K) * gate[:, None, :]
    ref = (v * torch.sigmoid(v)).view(M, K) @ w.t()
    y = torch.empty((M, ops.cpad(N)), dtype=dtype, device=DEV)
    ops.pw_gemm(padc(x, Kp).to(DEV, dtype).contiguous(), w.to(DEV), y, M=M, K=K, N=N, w_sn=K, w_sk=1,
                dtype=ops.dt_code(dtype), pro_mode=ops.PRO_BN_SE_SWISH, pro_p=torch.cat([padc(scale, Kp), padc(shift, Kp)]).to(DEV),
                pro_gate=padc(gate, Kp).to(DEV).contiguous(), rows_per_sample=rows)
    close(y[:, :N], ref, dtype, "y", scale=ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("res_mode", [0, 1])
def test_wide_affine2_transposed_add(dtype, res_mode):
    _need_gpu()
    from change3d_amd import ops
    BT, H, W = 3, 6, 8
    M, K, N = BT * H * W, 432, 96 if res_mode else 192     # conv_a data gradient of res5 (block 0: back to 96 channels)
    Kp = ops.cpad(K)
    g, a = q(rnd((M, K), 8), dtype), q(rnd((M, K), 9), dtype)
    A, Bc, Cc = rnd((K,), 10), rnd((K,), 11, 0.1), rnd((K,), 12, 0.1)
    wt = rnd((K, N), 13, 0.1)     # conv weight [out=K][in=N]
    if res_mode == 0:
        res = q(rnd((M, N), 14), dtype)
        full = res
    else:
        res = q(rnd((BT, H // 2, W // 2, N), 14), dtype)
        full = torch.zeros(BT, H, W, N)
        full[:, ::2, ::2] = res
        full = full.view(M, N)
    ref = (A * g + Bc + Cc * a) @ wt + full
    y = torch.empty((M, ops.cpad(N)), dtype=dtype, device=DEV)
    ops.pw_gemm(padc(g, Kp).to(DEV, dtype).contiguous(), wt.to(DEV), y, M=M, K=K, N=N, w_sn=1, w_sk=N,
                dtype=ops.dt_code(dtype), x2=padc(a, Kp).to(DEV, dtype).contiguous(), pro_mode=ops.PRO_AFFINE2,
                pro_p=torch.cat([padc(A, Kp), padc(Bc, Kp), padc(Cc, Kp)]).to(DEV), epi_mode=ops.EPI_ADD,
                e1=res.to(DEV, dtype).contiguous(), res_mode=res_mode, H=H, W=W)
    close(y[:, :N], ref, dtype, "y", scale=ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows", [48, 64, 176])
def test_wide_swish_se_bwd_epilogue(dtype, rows):
    _need_gpu()
    from change3d_amd import ops
    B, K, N = 3, 192, 432
    M = B * rows
    Np = ops.cpad(N)
    g = q(rnd((M, K), 20), dtype)
    w = rnd((K, N), 21, 0.1)  # conv_c weight [out=K=192][in=N=432]; data-grad = g @ w
    b = q(rnd((M, N), 22), dtype)
    scale, shift = rnd((N,), 23).abs() + 0.5, rnd((N,), 24, 0.3)
    gate = torch.sigmoid(rnd((B, N), 25))
    dsb = g @ w
    pb = b * scale + shift
    gg = gate.repeat_interleave(rows, 0)
    qv = gg * pb
    sg = torch.sigmoid(qv)
    dq = dsb * sg * (1 + qv * (1 - sg))
    t1 = dq * gg
    y = torch.empty((M, Np), dtype=dtype, device=DEV)
    nc3 = torch.zeros(B * Np * 3, dtype=torch.float64, device=DEV)
    mean, rstd = rnd((N,), 26, 0.5), rnd((N,), 27).abs() + 0.5
    ident = torch.cat([torch.ones(K), torch.zeros(K), torch.zeros(K)]).to(DEV)
    gd = g.to(DEV, dtype).contiguous()
    ops.pw_gemm(gd, w.to(DEV), y, M=M, K=K, N=N, w_sn=1, w_sk=N, dtype=ops.dt_code(dtype), x2=gd,
                pro_mode=ops.PRO_AFFINE2, pro_p=ident, epi_mode=ops.EPI_SWISH_SE_BWD, e1=padc(b, Np).to(DEV, dtype).contiguous(),
                epi_p=torch.cat([padc(scale, Np), padc(shift, Np)]).to(DEV), epi_gate=padc(gate, Np).to(DEV).contiguous(),
                epi_q=torch.cat([padc(mean, Np), padc(rstd, Np)]).to(DEV), stats=nc3, rows_per_sample=rows)
    close(y[:, :N], t1, dtype, "t1", scale=t1.abs().max().item())
    s = nc3.cpu().view(B, Np, 3)[:, :N]
    t1q = y[:, :N].float().cpu()
    ref0 = (dq * pb).view(B, rows, N).sum(1).double()
    ref1 = t1q.view(B, rows, N).sum(1).double()
    ref2 = (t1q * ((b - mean) * rstd)).view(B, rows, N).sum(1).double()
    rt = 2e-4 if dtype == torch.float32 else 3e-2
    assert torch.allclose(s[..., 0], ref0, rtol=rt, atol=rt * ref0.abs().max().item()), "sum dq*pb"
    assert torch.allclose(s[..., 1], ref1, rtol=2e-4, atol=2e-3), "sum t1"
    assert torch.allclose(s[..., 2], ref2, rtol=2e-4, atol=2e-3), "sum t1*bhat"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K,N", [(432, 192), (192, 432), (96, 432), (192, 576)])
def test_wide_wgrad(dtype, K, N):
    _need_gpu()
    from change3d_amd import ops
    B, rows = 3, 150
    M = B * rows
    Kp, Np = ops.cpad(K), ops.cpad(N)
    p, p2 = q(rnd((M, N), 30), dtype), q(rnd((M, N), 31), dtype)
    A, Bc, Cc = rnd((N,), 32), rnd((N,), 33, 0.1), rnd((N,), 34, 0.1)
    x = q(rnd((M, K), 35), dtype)
    scale, shift = rnd((K,), 36).abs() + 0.5, rnd((K,), 37, 0.3)
    gate = torch.sigmoid(rnd((B, K), 38))
    P = A * p + Bc + Cc * p2
    v = (x * scale + shift) * gate.repeat_interleave(rows, 0)
    Q = v * torch.sigmoid(v)
    ref = P.t() @ Q + 1.0
    dw = torch.ones((N, K), dtype=torch.float32, device=DEV)  # accumulate semantics (+=)
    ops.pw_wgrad(padc(p, Np).to(DEV, dtype).contiguous(), padc(x, Kp).to(DEV, dtype).contiguous(), dw, M=M, K=K, N=N,
                 dw_sn=K, dw_sk=1, dtype=ops.dt_code(dtype), p2=padc(p2, Np).to(DEV, dtype).contiguous(),
                 p_coef=torch.cat([padc(A, Np), padc(Bc, Np), padc(Cc, Np)]).to(DEV), q_mode=ops.PRO_BN_SE_SWISH,
                 q_ss=torch.cat([padc(scale, Kp), padc(shift, Kp)]).to(DEV),
                 q_gate=padc(gate, Kp).to(DEV).contiguous(), rows_per_sample=rows)
    close(dw, ref, dtype, "dW", scale=ref.abs().max().item())
    # plain operands (linear layer): dW = dY^T X
    dw2 = torch.zeros((N, K), dtype=torch.float32, device=DEV)
    ops.pw_wgrad(padc(p, Np).to(DEV, dtype).contiguous(), padc(x, Kp).to(DEV, dtype).contiguous(), dw2, M=M, K=K, N=N,
                 dw_sn=K, dw_sk=1, dtype=ops.dt_code(dtype))
    ref2 = p.t() @ x
    close(dw2, ref2, dtype, "dW plain", scale=ref2.abs().max().item())
