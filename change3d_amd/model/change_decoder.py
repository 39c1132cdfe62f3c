"""MI355X-native mirror of the reference's `model/change_decoder.py` (reference
model/change_decoder.py:10-81): same constructor `ChangeDecoder(args, in_dim, has_sigmoid)`,
same `forward(f: List[Tensor]) -> Tensor`, same submodule names (`up_c4/up_c3/up_c2/up_c1`
Sequentials of nn.Conv2d / nn.ConvTranspose2d, so `weight_init` (reference
model/utils.py:20-82) and the state-dict keys behave identically).  The nn children are
parameter holders; the computation is HIP kernels on channels-last feature maps:

    c3f = c3 + convT(up_c4)(conv1x1(c4))  ... -> pred = [sigmoid](conv3x3(c1f))

Inputs are the strided `x[:, :, k]` frame views the encoder hands over (reference
model/trainer.py:136-139); they are read in place (no gather copy).
"""
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..ops import cpad


def _frame_view(t):
    """(data_ptr, batch_stride_in_elements, tensor_kept_alive) of a logical [B,C,H,W] tensor whose
    memory is NHWC per sample (e.g. one frame of an NDHWC tensor)."""
    B, C, H, W = t.shape
    if not (t.stride(1) == 1 and t.stride(3) == C and t.stride(2) == W * C):
        t = t.contiguous(memory_format=torch.channels_last)
    return t.data_ptr(), t.stride(0), t


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c1, c2, c3, c4, anchor, dec):
        ops.require_gpu(c1, "decoder input")
        act = c1.dtype
        dt = ops.dt_code(act)
        dev = c1.device
        feats = [c1.detach(), c2.detach(), c3.detach(), c4.detach()]
        B = c1.shape[0]
        ups = [dec.up_c2, dec.up_c3, dec.up_c4]  # produces level 1, 2, 3 from level 2, 3, 4
        # top-down: cur = c4 view
        ptr, bstride, keep = _frame_view(feats[3])
        keepalive = [keep]
        cur_dense = None
        saved = []
        for lvl in (3, 2, 1):  # input level index in feats (c4, then c3f, c2f)
            conv, convt = ups[lvl - 1][0], ups[lvl - 1][1]
            Cin, Cout = conv.weight.shape[1], conv.weight.shape[0]
            h, w = feats[lvl].shape[2], feats[lvl].shape[3]
            M = B * h * w
            t = torch.empty((M, cpad(Cout)), dtype=act, device=dev)
            if cur_dense is None:
                ops.pw_gemm(None, conv.weight, t, M=M, K=Cin, N=Cout, w_sn=Cin, w_sk=1, dtype=dt,
                            row_mode=ops.ROWS_FRAME, rpg=h * w, gstride=bstride, x_ptr=ptr)
            else:
                ops.pw_gemm(cur_dense, conv.weight, t, M=M, K=Cin, N=Cout, w_sn=Cin, w_sk=1, dtype=dt)
            sptr, sbs, skeep = _frame_view(feats[lvl - 1])
            keepalive.append(skeep)
            out = torch.empty((B, 2 * h, 2 * w, Cout), dtype=act, device=dev)
            ops.convT_fwd(t, convt.weight, convt.bias, sptr, sbs, out, B, h, w, Cout, dt)
            saved.append((cur_dense, ptr, bstride, t, h, w, Cin, Cout))
            cur_dense = out
        H, W = cur_dense.shape[1], cur_dense.shape[2]
        head = dec.up_c1[0]
        NC = head.weight.shape[0]
        pred = torch.empty((B, NC, H, W), dtype=torch.float32, device=dev)
        ops.head_fwd(cur_dense, head.weight, pred, B, H, W, head.weight.shape[1], NC, dec.has_sigmoid, dt)
        ctx.dec, ctx.saved, ctx.c1f, ctx.pred, ctx.keepalive = dec, saved, cur_dense, pred, keepalive
        ctx.act = act
        return pred

    @staticmethod
    def backward(ctx, dpred):
        dec, saved, c1f, pred, act = ctx.dec, ctx.saved, ctx.c1f, ctx.pred, ctx.act
        dt = ops.dt_code(act)
        dev = dpred.device
        B, H, W, C1 = c1f.shape
        head = dec.up_c1[0]
        NC = head.weight.shape[0]
        dcur = torch.empty_like(c1f)
        ops.head_bwd(dpred.contiguous().float(), pred, c1f, head.weight, dcur, ops.grad_of(head.weight), B, H, W, C1,
                     NC, dec.has_sigmoid, dt)
        ups = [dec.up_c2, dec.up_c3, dec.up_c4]
        grads = [None, None, None, None]
        grads[0] = dcur.permute(0, 3, 1, 2)  # d c1 = d c1f (skip add)
        for lvl, (x_dense, ptr, bstride, t, h, w, Cin, Cout) in zip((1, 2, 3), reversed(saved)):
            conv, convt = ups[lvl - 1][0], ups[lvl - 1][1]
            M = B * h * w
            # dcur is d(out) at [B,2h,2w,Cout]
            gb, gw = ops.grad_of(convt.bias), ops.grad_of(convt.weight)  # gw: [Cin=Cout][Cout][4][4]
            dt_ = torch.empty((M, cpad(Cout)), dtype=act, device=dev)
            ops.convT_bwd_data(dcur, convt.weight, dt_, B, h, w, Cout, dt)

            def convt_param_grads(dcur=dcur, t=t, gb=gb, gw=gw, M=M, h=h, w=w, Cout=Cout):
                ops.col_sum(dcur, gb, B * 4 * h * w, Cout, dt)
                # the 4x4 taps (ky, kx) in ONE launch: tap t reads dcur at (2i + ky - 1, 2j + kx - 1) and
                # accumulates at gw[ci][co][ky][kx] (16 launches of ~20-50 us each before)
                if dt == ops.DT_BF16 and ops.CONVT_MFMA and Cout in (24, 48):
                    ops.convT_wgrad(t, dcur, gw, B, h, w, Cout, dt)   # one MFMA pass over t and dcur for all 16 taps
                else:
                    ops.pw_wgrad(t, dcur, gw, M=M, K=Cout, N=Cout, dw_sn=Cout * 16, dw_sk=16, dtype=dt,
                                 row_mode=ops.ROWS_S2SHIFT, H=2 * h, W=2 * w, taps=16, dw_tap_stride=1)

            ops.side_run(convt_param_grads, dcur, t)   # leaves of the backward graph: side stream
            dx = torch.empty((B, h, w, cpad(Cin)), dtype=act, device=dev)
            ops.pw_gemm(dt_, conv.weight, dx, M=M, K=Cout, N=Cin, w_sn=1, w_sk=Cin, dtype=dt)
            gc = ops.grad_of(conv.weight)
            if x_dense is None:
                ops.side_run(lambda dt_=dt_, gc=gc, M=M, Cin=Cin, Cout=Cout, h=h, w=w, bstride=bstride, ptr=ptr:
                             ops.pw_wgrad(dt_, None, gc, M=M, K=Cin, N=Cout, dw_sn=Cin, dw_sk=1, dtype=dt,
                                          row_mode=ops.ROWS_FRAME, rpg=h * w, gstride=bstride, q_ptr=ptr), dt_)
            else:
                ops.side_run(lambda dt_=dt_, x_dense=x_dense, gc=gc, M=M, Cin=Cin, Cout=Cout:
                             ops.pw_wgrad(dt_, x_dense, gc, M=M, K=Cin, N=Cout, dw_sn=Cin, dw_sk=1, dtype=dt), dt_, x_dense)
            grads[lvl] = dx.permute(0, 3, 1, 2)  # d c_{lvl+1}: skip grad at that level == d(c_f) == dx
            dcur = dx
        return grads[0], grads[1], grads[2], grads[3], None, None


class ChangeDecoder(nn.Module):
    """Decoder network that upsamples feature maps and produces final predictions."""

    def __init__(self, args, in_dim: List[int] = [64, 128, 256, 384], has_sigmoid: bool = False) -> None:
        super().__init__()
        self.has_sigmoid = has_sigmoid
        c1, c2, c3, c4 = in_dim
        if (c1, c2, c3, c4) != (24, 24, 48, 96):
            raise NotImplementedError("decoder kernels are specialised for Change3D's embed dims [24,24,48,96]")

        def up(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False),
                                 nn.ConvTranspose2d(cout, cout, kernel_size=4, stride=2, padding=1))

        self.up_c4, self.up_c3, self.up_c2 = up(c4, c3), up(c3, c2), up(c2, c1)
        num_class = 1 if has_sigmoid else args.num_class
        if num_class > 8:
            raise NotImplementedError("head kernel supports num_class <= 8")
        self.up_c1 = nn.Sequential(nn.Conv2d(c1, num_class, kernel_size=3, stride=1, padding=1, bias=False))

    def forward(self, f: List[torch.Tensor]) -> torch.Tensor:
        c1, c2, c3, c4 = f
        return _DecoderFn.apply(c1, c2, c3, c4, self.up_c1[0].weight, self)
