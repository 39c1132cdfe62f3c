"""MI355X-native mirror of the reference's `model/x3d.py` (reference model/x3d.py:543-744).

`create_x3d(**kw)` keeps the reference's keyword-only signature and returns an `nn.Module`
whose `.blocks` is an indexable list of 6 callables on logical-NCDHW tensors
(`blocks[i](x)` is how reference model/trainer.py:122,130 drives it) and whose state-dict
keys are exactly the reference's 1 141 keys (`blocks.0.conv.conv_t.weight`, ...,
`blocks.S.res_blocks.J.branch2.norm_b.1.block.0.weight`, `blocks.5.proj.bias`), so
`X3D_L.pyth['model_state']` still strict-loads (reference model/trainer.py:43-45).

The nn.Conv3d / nn.BatchNorm3d children are PARAMETER HOLDERS only: every forward/backward
below is a sequence of hand-written gfx950 kernels from libchange3d_hip.so (via ../ops.py).
Only the X3D-L configuration Change3D instantiates is supported (reference
model/trainer.py:40: `create_x3d(input_clip_length=3, depth_factor=5.0)`); other widths
raise NotImplementedError.  Activations between kernels are channels-last
[B][T][H][W][C] in `act_dtype` (float32 = parity path, bfloat16 = throughput path).
"""
import math

import os

import torch
import torch.nn as nn

from .. import ops
from ..ops import cpad

BN_EPS, BN_MOM = 1e-5, 0.1


# ------------------------------------------------------------------ helpers (restated utils)
def round_width(width, multiplier, min_width=8, divisor=8, ceil=False):
    """pytorchvideo.layers.utils.round_width as used at reference model/x3d.py:197,657,675-683."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    if ceil:
        out = max(min_width, int(math.ceil(width / divisor)) * divisor)
    else:
        out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def round_repeats(repeats, multiplier):
    return repeats if not multiplier else int(math.ceil(multiplier * repeats))


def to_ndhwc(x):
    """logical NCDHW (any strides) -> contiguous [B,T,H,W,C] tensor (no copy if channels_last_3d)."""
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_logical(y):
    """[B,T,H,W,C] contiguous -> logical NCDHW view (channels_last_3d strides)."""
    return y.permute(0, 4, 1, 2, 3)


class _Holder(nn.Module):
    """Plain container; keeps attribute names identical to the reference module tree."""


def _f32(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


# ---------------------------------------------------------------------------------- stem
def _stem_forward(ctx, xin, stem):
    B, Ci, T, H, W = xin.shape
    dev, dt = xin.device, ops.dt_code(stem.act_dtype)
    training = stem.training
    conv_s, conv_t = stem.conv.conv_t, stem.conv.conv_xy  # names swapped upstream (x3d.py:87-92)
    C = conv_s.weight.shape[0]
    sums = torch.zeros(2 * C, dtype=torch.float64, device=dev) if training else None
    u = torch.empty((B, T, H, W, C), dtype=stem.act_dtype, device=dev)
    ops.stem_fwd(xin, conv_s.weight, conv_t.weight, u, sums, B, T, H, W, dt)
    ss, mr = _f32(2 * cpad(C), dev), _f32(2 * cpad(C), dev)
    ops.bn_finalize(sums, B * T * H * W, stem.norm, C, ss, mr, training)
    y = torch.empty_like(u)
    ops.block_out_fwd(u, ss, None, None, ops.SC_NONE, y, B * T * H * W, cpad(C), dt)
    ctx.stem, ctx.saved = stem, (xin, u, y, mr)
    return to_logical(y)


def _stem_backward(ctx, dy, frames_grad=None):
    """Returns the per-sample input gradient (or None); with `frames_grad` = (tensor [3][K][H][W], t_first, K) the
    batch-summed gradient of those frames is accumulated into it instead."""
    stem = ctx.stem
    xin, u, y, mr = ctx.saved
    B, _, T, H, W = xin.shape
    dev, dt = xin.device, ops.dt_code(stem.act_dtype)
    conv_s, conv_t = stem.conv.conv_t, stem.conv.conv_xy
    C = conv_s.weight.shape[0]
    M = B * T * H * W
    dyc = to_ndhwc(dy).to(stem.act_dtype)
    g = torch.empty_like(u)
    dsums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    ops.block_out_bwd(dyc, y, u, None, g, mr, None, dsums, None, M, C, dt)
    coef = _f32(3 * cpad(C), dev)
    ops.bn_bwd_coef(dsums, M, stem.norm, mr, C, coef)
    dv = torch.empty_like(u)
    ops.stem_bwd_dv(xin, conv_s.weight, conv_t.weight, g, u, coef, dv, ops.grad_of(conv_t.weight), B, T, H, W, dt)
    dx = None
    if frames_grad is not None:
        gp, t0, nf = frames_grad
        ops.stem_bwd_wx(xin, conv_s.weight, dv, ops.grad_of(conv_s.weight), gp, B, T, H, W, t0, nf, False, dt)
    elif ctx.x_needs_grad:
        t0, nf = stem.grad_frames if stem.grad_frames is not None else (0, T)
        dx = torch.zeros_like(xin)
        ops.stem_bwd_wx(xin, conv_s.weight, dv, ops.grad_of(conv_s.weight), dx, B, T, H, W, t0, nf, True, dt)
    else:
        ops.stem_bwd_wx(xin, conv_s.weight, dv, ops.grad_of(conv_s.weight), None, B, T, H, W, 0, 0, False, dt)
    return dx


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, stem):
        ops.require_gpu(x, "stem input")
        if x.shape[1] != 3:
            raise NotImplementedError("stem kernel is specialised for 3 input channels")
        ctx.x_needs_grad = x.requires_grad
        return _stem_forward(ctx, x.detach().contiguous().float(), stem)

    @staticmethod
    def backward(ctx, dy):
        return _stem_backward(ctx, dy), None, None


class _ClipStemFn(torch.autograd.Function):
    """`stem(cat([pre, perception_frames.expand(B), post], dim=2))` (reference model/trainer.py:155-162 + :128-130 for
    block 0) as ONE function: the clip is assembled by a HIP copy kernel, and in backward the stem's input-gradient
    kernel writes the BATCH-SUMMED gradient of the perception frames straight into their `.grad` (no full-size
    zero-filled dx, no expand-backward reduction)."""

    @staticmethod
    def forward(ctx, pre, post, frames, anchor, stem):
        ops.require_gpu(pre, "encoder input")
        B, Ci, H, W = pre.shape
        K = frames.shape[2]
        if Ci != 3 or tuple(frames.shape) != (1, 3, K, H, W) or tuple(post.shape) != tuple(pre.shape):
            raise NotImplementedError("clip assembly expects (B,3,H,W) images and (1,3,K,H,W) perception frames")
        clip = torch.empty((B, 3, K + 2, H, W), dtype=torch.float32, device=pre.device)
        ops.build_clip(pre.detach().contiguous().float(), post.detach().contiguous().float(),
                       frames.detach().contiguous().float(), clip, B, K, H, W)
        ctx.frames, ctx.K, ctx.x_needs_grad = frames, K, False
        ctx.need_frames = ctx.needs_input_grad[2]
        return _stem_forward(ctx, clip, stem)

    @staticmethod
    def backward(ctx, dy):
        gp = None
        if ctx.need_frames:
            gp = torch.zeros(ctx.frames.shape, dtype=torch.float32, device=dy.device)
        _stem_backward(ctx, dy, (gp, 1, ctx.K) if gp is not None else None)
        return None, None, gp, None, None


class X3DStem(nn.Module):
    """`blocks[0]`: Conv2plus1d(conv_t = spatial 1x3x3, conv_xy = temporal 5x1x1 dw) -> BN -> ReLU
    (reference model/x3d.py:23-106).  `grad_frames=(t_first, n)` limits the input gradient to
    those frames (Change3D only needs the perception frames)."""

    def __init__(self, cin, cout, ksize, stride, act_dtype):
        super().__init__()
        if tuple(ksize) != (5, 3, 3) or tuple(stride) != (1, 1, 1) or cout != 24:
            raise NotImplementedError("stem kernels are specialised for k=(5,3,3), stride 1, 24 channels")
        self.conv = _Holder()
        self.conv.conv_t = nn.Conv3d(cin, cout, (1, 3, 3), stride=(1, 1, 1), padding=(0, 1, 1), bias=False)
        self.conv.conv_xy = nn.Conv3d(cout, cout, (5, 1, 1), stride=(1, 1, 1), padding=(2, 0, 0), bias=False,
                                      groups=cout)
        self.norm = nn.BatchNorm3d(cout, eps=BN_EPS, momentum=BN_MOM)
        self.act_dtype = act_dtype
        self.grad_frames = None

    def forward(self, x):
        return _StemFn.apply(x, self.norm.weight, self)


# ------------------------------------------------------------------------------ res stage
class X3DBottleneck(_Holder):
    pass


class X3DResBlock(nn.Module):
    """Parameter holder for one residual block (reference model/x3d.py:235-328, 109-232)."""

    def __init__(self, cin, cinner, cout, stride, use_se, se_ratio):
        super().__init__()
        self.cin, self.cinner, self.cout, self.stride, self.use_se = cin, cinner, cout, stride, use_se
        need_conv = cin != cout or stride > 1
        self.branch1_conv = (nn.Conv3d(cin, cout, (1, 1, 1), stride=(1, stride, stride), bias=False)
                             if need_conv else None)
        self.branch1_norm = nn.BatchNorm3d(cout) if cin != cout else None
        b2 = X3DBottleneck()
        b2.conv_a = nn.Conv3d(cin, cinner, (1, 1, 1), bias=False)
        b2.norm_a = nn.BatchNorm3d(cinner, eps=BN_EPS, momentum=BN_MOM)
        b2.conv_b = nn.Conv3d(cinner, cinner, (3, 3, 3), stride=(1, stride, stride), padding=(1, 1, 1),
                              bias=False, groups=cinner)
        if use_se:
            se = _Holder()
            cr = round_width(cinner, se_ratio)
            se.block = nn.Sequential(nn.Conv3d(cinner, cr, 1, bias=True), nn.ReLU(),
                                     nn.Conv3d(cr, cinner, 1, bias=True), nn.Sigmoid())
        else:
            se = nn.Identity()
        b2.norm_b = nn.Sequential(nn.BatchNorm3d(cinner, eps=BN_EPS, momentum=BN_MOM), se)
        b2.conv_c = nn.Conv3d(cinner, cout, (1, 1, 1), bias=False)
        b2.norm_c = nn.BatchNorm3d(cout, eps=BN_EPS, momentum=BN_MOM)
        self.branch2 = b2


class _StageFn(torch.autograd.Function):
    """One residual stage = ONE C call forward and ONE backward (`c3d_stage_fwd` / `c3d_stage_bwd`,
    csrc/stage_driver.hip): the per-block launch sequence runs in C++ over a single workspace; weight gradients go
    to the driver's side stream.  This is the only launch path: per-kernel profiles are taken by the driver itself
    (`ops.profile_begin`, `c3d_prof_begin`)."""

    @staticmethod
    def forward(ctx, x, anchor, stage, grad_mode):
        ops.require_gpu(x, "stage input")
        B, C, T, H, W = x.shape
        act = stage.act_dtype
        xin = to_ndhwc(x.detach()).to(act)
        if xin.shape[-1] != cpad(C):
            raise NotImplementedError("stage input channels must be a multiple of 8")
        bind = stage.binding()
        # needs_input_grad reports requires_grad of the inputs even under torch.no_grad(); the caller's grad mode
        # (grad mode is always off inside Function.forward) is passed in explicitly
        keep = grad_mode and any(ctx.needs_input_grad)
        ctx.eval_graph = False
        if keep and not stage.training:
            # eval()-mode stage called with grad mode on (`model.eval(); model(x)` without torch.no_grad(), as the reference
            # allows): the forward runs -- on the inference path, nothing saved -- and only backward() through it refuses:
            # eval-mode BatchNorm's backward is dx = gamma*rstd*g, not the batch-statistics form the backward kernels implement
            keep, ctx.eval_graph = False, True
        bn0 = stage.res_blocks[0].branch2.norm_a
        bind.desc.flags = int(stage.driver_flags)
        bind.refresh(B, T, H, W, ops.dt_code(act), stage.training, float(bn0.momentum), float(bn0.eps), with_grads=False)
        ws_bytes, _, y_bytes, _ = bind.sizes()
        last = stage.res_blocks[-1]
        s0 = stage.res_blocks[0].stride
        Ho, Wo = (H - 1) // s0 + 1, (W - 1) // s0 + 1
        y = torch.empty((B, T, Ho, Wo, cpad(last.cout)), dtype=act, device=x.device)
        assert y.numel() * y.element_size() == y_bytes
        if not stage.training and not keep and stage.fold_bn_eval:
            # inference: BatchNorm folded into the conv weights (reference scripts/train_BCD.py:92-154 val())
            fold_bytes, ws_eval = ops.stage_fold_sizes(bind)
            if stage._fold is None or stage._fold.numel() != fold_bytes or stage._fold.device != x.device:
                stage._fold, stage._fold_valid = torch.empty(fold_bytes, dtype=torch.uint8, device=x.device), False
            if not stage._fold_valid or stage._fold_version != ops.weights_version():
                ops.stage_fold_bn(bind, stage._fold)
                stage._fold_valid, stage._fold_version = True, ops.weights_version()
            ws = torch.empty(ws_eval, dtype=torch.uint8, device=x.device)
            ops.stage_fwd_folded(bind, stage._fold, xin, ws, y)
            return to_logical(y)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        ops.stage_fwd(bind, xin, ws, y)
        if keep:
            ctx.stage, ctx.saved_ws, ctx.xin, ctx.y, ctx.x_dtype, ctx.dims = stage, ws, xin, y, x.dtype, (B, T, H, W)
        else:
            ctx.saved_ws = None
        return to_logical(y)

    @staticmethod
    def backward(ctx, dy):
        if ctx.eval_graph:
            raise NotImplementedError("gradients through an eval()-mode residual stage are not implemented (the backward "
                                      "kernels implement train-mode BatchNorm); call .train() before the forward pass")
        stage, ws, xin, y = ctx.stage, ctx.saved_ws, ctx.xin, ctx.y
        if ws is None:
            raise RuntimeError("Trying to backward through a residual stage a second time: its saved activations were "
                               "released by the first backward pass (retain_graph is not supported by the stage driver)")
        B, T, H, W = ctx.dims
        act = stage.act_dtype
        dyc = to_ndhwc(dy).to(act)
        bind = stage.binding()
        bn0 = stage.res_blocks[0].branch2.norm_a
        bind.desc.flags = int(stage.driver_flags)
        bind.refresh(B, T, H, W, ops.dt_code(act), True, float(bn0.momentum), float(bn0.eps), with_grads=True)
        _, wb_bytes, _, dx_bytes = bind.sizes()
        wb = torch.empty(wb_bytes, dtype=torch.uint8, device=dy.device)
        dx = torch.empty(xin.shape, dtype=act, device=dy.device)
        assert dx.numel() * dx.element_size() == dx_bytes
        ops.stage_bwd(bind, xin, y, dyc, ws, wb, dx)
        # the driver's side stream may still read these: they are released at the next full side_join()
        # (queued here as the autograd end-of-pass callback, or right now outside a backward pass)
        ops.keep_until_join(ws, wb, xin, y, dyc)
        ctx.saved_ws = ctx.xin = ctx.y = None
        if stage.post_backward is not None:  # data-parallel hook: this stage's grads must be final
            ops.side_join()
            stage.post_backward()
        return to_logical(dx).to(ctx.x_dtype), None, None, None


class X3DResStage(nn.Module):
    """`blocks[1..4]` (reference model/x3d.py:331-412)."""

    def __init__(self, depth, cin, cinner, cout, stride, se_ratio, act_dtype):
        super().__init__()
        self.res_blocks = nn.ModuleList([
            X3DResBlock(cin if i == 0 else cout, cinner, cout, stride if i == 0 else 1,
                        use_se=bool((i + 1) % 2) and se_ratio > 0, se_ratio=se_ratio) for i in range(depth)])
        self.act_dtype = act_dtype
        self.post_backward = None
        self._binding = None
        # eval / no-grad forward with BatchNorm folded into the conv weights.  The folded copy is rebuilt lazily
        # after every train()/eval() switch, load_state_dict(), .to(), and whenever the library's own in-place
        # writers ran since it was made (FusedAdam.launch and broadcast_module_state bump ops.weights_version());
        # code that edits parameters or running statistics by hand while the module stays in eval mode must call
        # invalidate_folded_bn().
        self.fold_bn_eval = os.environ.get("C3D_FOLD_BN", "1") != "0"
        self._fold, self._fold_valid, self._fold_version = None, False, -1
        self.driver_flags = 0   # c3d_stage_desc.flags (ops.STAGE_*): the unfused launch sequences, for parity tests

    def invalidate_folded_bn(self):
        self._fold_valid = False

    def train(self, mode=True):
        self._fold_valid = False
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._fold_valid = False
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._fold_valid = False
        self._binding = None      # Module._apply replaces buffer tensors: re-resolve the bound objects
        return super()._apply(fn, *args, **kwargs)

    def binding(self):
        if self._binding is None:
            self._binding = ops.StageBinding(self)
        return self._binding

    def forward(self, x):
        return _StageFn.apply(x, self.res_blocks[0].branch2.conv_a.weight, self, torch.is_grad_enabled())


class X3DHead(nn.Module):
    """`blocks[5]`: constructed and strict-loaded by the reference, never executed by any
    Change3D path (reference model/trainer.py:128 runs range(4), :121 range(5)); kept as a
    parameter holder for state-dict / parameters() parity (reference model/x3d.py:415-540)."""

    def __init__(self, cin, cinner, cout, num_classes, dropout_rate):
        super().__init__()
        self.pool = _Holder()
        self.pool.pre_conv = nn.Conv3d(cin, cinner, (1, 1, 1), bias=False)
        self.pool.pre_norm = nn.BatchNorm3d(cinner, eps=BN_EPS, momentum=BN_MOM)
        self.pool.post_conv = nn.Conv3d(cinner, cout, (1, 1, 1), bias=False)
        self.proj = nn.Linear(cout, num_classes, bias=True)
        self.dropout_rate = dropout_rate

    def forward(self, x):
        raise NotImplementedError("the X3D classification head is never executed by Change3D; "
                                  "it exists for state-dict compatibility only")


class X3DNet(nn.Module):
    def __init__(self, blocks, act_dtype):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        self.act_dtype = act_dtype
        _init_net_weights(self)

    def set_act_dtype(self, dtype):
        self.act_dtype = dtype
        for m in self.modules():
            if hasattr(m, "act_dtype"):
                m.act_dtype = dtype
        return self

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


def _init_net_weights(model, fc_init_std=0.01):
    """pytorchvideo `init_net_weights` (called by its `Net.__init__`): kaiming-normal fan_out for
    convs, BN weight 1 / bias 0, Linear N(0, 0.01)."""
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm3d):
            m.weight.data.fill_(1.0)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=fc_init_std)
            m.bias.data.zero_()


class Swish(nn.Module):
    """`pytorchvideo.layers.swish.Swish` as the reference imports it (reference model/x3d.py:13-20): the
    default `inner_act` of the bottleneck.  Inside `create_x3d*` it is a MARKER (Swish is fused into the
    pointwise-GEMM prologue/epilogue kernels); standalone it evaluates x * sigmoid(x)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


def _check_norm_act(norm, norm_eps, norm_momentum, activation, what):
    if norm is not nn.BatchNorm3d or activation is not nn.ReLU or norm_eps != BN_EPS or norm_momentum != BN_MOM:
        raise NotImplementedError(f"{what}: only BatchNorm3d(eps=1e-5, momentum=0.1) + ReLU is implemented as HIP kernels")


def create_x3d_stem(*, in_channels, out_channels, conv_kernel_size=(5, 3, 3), conv_stride=(1, 2, 2),
                    conv_padding=(2, 1, 1), norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1,
                    activation=nn.ReLU, act_dtype=torch.float32):
    """reference model/x3d.py:23-106 (same keywords).  Change3D calls it with stride (1,1,1)
    (reference model/x3d.py:564); that is the configuration the stem kernels implement."""
    _check_norm_act(norm, norm_eps, norm_momentum, activation, "create_x3d_stem")
    if tuple(conv_padding) != tuple(k // 2 for k in conv_kernel_size):
        raise NotImplementedError("stem kernels use 'same' padding")
    return X3DStem(in_channels, out_channels, conv_kernel_size, conv_stride, act_dtype)


def create_x3d_bottleneck_block(*, dim_in, dim_inner, dim_out, conv_kernel_size=(3, 3, 3), conv_stride=(1, 2, 2),
                                norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, se_ratio=0.0625,
                                activation=nn.ReLU, inner_act=Swish):
    """reference model/x3d.py:109-232.  Returns the `branch2` PARAMETER HOLDER (conv_a/norm_a/conv_b/norm_b/
    conv_c/norm_c with the reference's key names); it is executed only as part of a residual stage
    (`create_x3d_res_stage`), whose kernels fuse across the block.  `se_ratio > 0` puts an SE block here, as
    the reference does (the stage passes 0 for odd block indices)."""
    _check_norm_act(norm, norm_eps, norm_momentum, activation, "create_x3d_bottleneck_block")
    if inner_act is not Swish or tuple(conv_kernel_size) != (3, 3, 3) or conv_stride[0] != 1 or conv_stride[1] != conv_stride[2]:
        raise NotImplementedError("bottleneck kernels: 3x3x3 depthwise, spatial stride 1 or 2, Swish inner activation")
    blk = X3DResBlock(dim_in, dim_inner, dim_out, conv_stride[1], use_se=se_ratio > 0.0, se_ratio=se_ratio)
    return blk.branch2


def create_x3d_res_block(*, dim_in, dim_inner, dim_out, bottleneck=create_x3d_bottleneck_block, use_shortcut=True,
                         conv_kernel_size=(3, 3, 3), conv_stride=(1, 2, 2), norm=nn.BatchNorm3d, norm_eps=1e-5,
                         norm_momentum=0.1, se_ratio=0.0625, activation=nn.ReLU, inner_act=Swish):
    """reference model/x3d.py:235-328: parameter holder for one residual block (branch1_conv / branch1_norm /
    branch2); executed inside a stage."""
    _check_norm_act(norm, norm_eps, norm_momentum, activation, "create_x3d_res_block")
    if bottleneck is not create_x3d_bottleneck_block or inner_act is not Swish or not use_shortcut \
            or tuple(conv_kernel_size) != (3, 3, 3) or conv_stride[0] != 1 or conv_stride[1] != conv_stride[2]:
        raise NotImplementedError("only the X3D residual block Change3D uses is implemented as HIP kernels")
    return X3DResBlock(dim_in, dim_inner, dim_out, conv_stride[1], use_se=se_ratio > 0.0, se_ratio=se_ratio)


def create_x3d_res_stage(*, depth, dim_in, dim_inner, dim_out, bottleneck=create_x3d_bottleneck_block,
                         conv_kernel_size=(3, 3, 3), conv_stride=(1, 2, 2), norm=nn.BatchNorm3d, norm_eps=1e-5,
                         norm_momentum=0.1, se_ratio=0.0625, activation=nn.ReLU, inner_act=Swish,
                         act_dtype=torch.float32):
    """reference model/x3d.py:331-412: `depth` blocks, stride on block 0 only, SE on even block indices."""
    _check_norm_act(norm, norm_eps, norm_momentum, activation, "create_x3d_res_stage")
    if bottleneck is not create_x3d_bottleneck_block or inner_act is not Swish \
            or tuple(conv_kernel_size) != (3, 3, 3) or conv_stride[0] != 1 or conv_stride[1] != conv_stride[2]:
        raise NotImplementedError("only the X3D residual stage Change3D uses is implemented as HIP kernels")
    return X3DResStage(depth, dim_in, dim_inner, dim_out, conv_stride[1], se_ratio, act_dtype)


def create_x3d_head(*, dim_in, dim_inner, dim_out, num_classes, pool_act=nn.ReLU, pool_kernel_size=(13, 5, 5),
                    norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, bn_lin5_on=False, dropout_rate=0.5,
                    activation=nn.Softmax, output_with_global_average=True):
    """reference model/x3d.py:415-540: parameter holder (never executed by any Change3D path)."""
    if bn_lin5_on:
        raise NotImplementedError("head_bn_lin5_on is not used by Change3D")
    return X3DHead(dim_in, dim_inner, dim_out, num_classes, dropout_rate)


def create_x3d(*, input_channel=3, input_clip_length=13, input_crop_size=160, model_num_class=400,
               dropout_rate=0.5, width_factor=2.0, depth_factor=2.2, norm=nn.BatchNorm3d, norm_eps=1e-5,
               norm_momentum=0.1, activation=nn.ReLU, stem_dim_in=12, stem_conv_kernel_size=(5, 3, 3),
               stem_conv_stride=(1, 1, 1), stage_conv_kernel_size=((3, 3, 3),) * 4,
               stage_spatial_stride=(2, 2, 2, 2), stage_temporal_stride=(1, 1, 1, 1),
               bottleneck=create_x3d_bottleneck_block, bottleneck_factor=2.25, se_ratio=0.0625, inner_act=Swish,
               head_dim_out=2048, head_pool_act=nn.ReLU, head_bn_lin5_on=False, head_activation=None,
               head_output_with_global_average=True, act_dtype=torch.float32):
    """Same keyword surface and defaults as reference model/x3d.py:543-584 (+ `act_dtype`, the activation
    storage type of the HIP path)."""
    if (norm is not nn.BatchNorm3d or activation is not nn.ReLU or norm_eps != BN_EPS or norm_momentum != BN_MOM
            or tuple(stage_spatial_stride) != (2, 2, 2, 2) or tuple(stage_temporal_stride) != (1, 1, 1, 1)
            or any(tuple(k) != (3, 3, 3) for k in stage_conv_kernel_size) or head_bn_lin5_on
            or head_activation is not None or bottleneck is not create_x3d_bottleneck_block or inner_act is not Swish):
        raise NotImplementedError("only the X3D configuration Change3D uses is implemented as HIP kernels")
    stem_out = round_width(stem_dim_in, width_factor)
    blocks = [create_x3d_stem(in_channels=input_channel, out_channels=stem_out, conv_kernel_size=stem_conv_kernel_size,
                              conv_stride=stem_conv_stride,
                              conv_padding=tuple(k // 2 for k in stem_conv_kernel_size), act_dtype=act_dtype)]
    dims = [stem_dim_in]
    for _ in range(3):
        dims.append(round_width(dims[-1], 2.0, divisor=8))
    cin = stem_out
    for i, d in enumerate((1, 2, 5, 3)):
        cout = round_width(dims[i], width_factor)
        cinner = int(bottleneck_factor * cout)
        blocks.append(create_x3d_res_stage(depth=round_repeats(d, depth_factor), dim_in=cin, dim_inner=cinner,
                                           dim_out=cout, conv_stride=(stage_temporal_stride[i], stage_spatial_stride[i],
                                                                      stage_spatial_stride[i]),
                                           se_ratio=se_ratio, act_dtype=act_dtype))
        cin = cout
    blocks.append(create_x3d_head(dim_in=cin, dim_inner=cinner, dim_out=head_dim_out, num_classes=model_num_class,
                                  dropout_rate=dropout_rate))
    return X3DNet(blocks, act_dtype)


def stage_saved_activations(y):
    """Debug / test access to what a residual stage kept for backward: for the stage output `y` (logical
    NCDHW tensor returned by `X3DResStage.forward` with grad enabled) a list with one dict per block holding the
    stored activation tensors `a`, `b`, `c`, `sc` ([rows, Cp] in the activation dtype; `sc` only where the
    shortcut has a BatchNorm), their BatchNorm (mean | rstd) vectors `mr_a`, `mr_b`, `mr_c`, `mr_sc` (f32
    [2][Cp]) and the real channel counts `C_a` ... -- tests recompute the statistics in f64 from these
    (tests/test_bf16_fullsize_gpu.py)."""
    fn = y.grad_fn
    if fn is None or not hasattr(fn, "stage"):
        raise ValueError("not the output of a residual stage evaluated with grad enabled")
    out = []
    if hasattr(fn, "saved_ws"):   # C++ stage driver: views into the forward workspace
        bind, ws, act = fn.stage.binding(), fn.saved_ws, fn.stage.act_dtype
        for i, blk in enumerate(fn.stage.res_blocks):
            rec = dict(C_a=blk.cinner, C_b=blk.cinner, C_c=blk.cout, C_sc=blk.cout)
            for name, cp in (("a", cpad(blk.cinner)), ("b", cpad(blk.cinner)), ("c", cpad(blk.cout)), ("sc", cpad(blk.cout))):
                if name == "sc" and blk.branch1_norm is None:
                    continue
                off, n = bind.saved(i, name)
                rec[name] = ws[off:off + n].view(act).view(-1, cp)
                off, n = bind.saved(i, "mr_" + name)
                rec["mr_" + name] = ws[off:off + n].view(torch.float32)
            out.append(rec)
        return out
    for blk, sv in zip(fn.stage.res_blocks, fn.saved):
        rec = dict(a=sv["a"], b=sv["b"], c=sv["c"], mr_a=sv["mr_a"], mr_b=sv["mr_b"], mr_c=sv["mr_c"],
                   C_a=blk.cinner, C_b=blk.cinner, C_c=blk.cout)
        if sv["mr_1"] is not None:
            rec.update(sc=sv["sc"], mr_sc=sv["mr_1"], C_sc=blk.cout)
        out.append(rec)
    return out
