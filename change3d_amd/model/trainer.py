"""MI355X-native mirror of the reference's `model/trainer.py` (reference model/trainer.py:20-306):
`Encoder(args, embed_dims)` and `Trainer(args)` with the same attributes, state-dict keys
(`encoder.x3d.*`, `encoder.perception_frames`, `encoder.fc.{i}.0.weight`, `decoder.*`) and
`update_bcd / update_scd / update_bda` methods, so a `scripts/train_BCD.py`-shaped driver
calls it unchanged.  Every tensor op on the path is a HIP kernel (x3d.py, change_decoder.py,
ops.py); `args.act_dtype` (optional, default float32) selects the activation storage type.
"""
from typing import Any, List

import torch
import torch.nn as nn

from .. import ops
from ..ops import cpad
from .change_decoder import ChangeDecoder
from .utils import weight_init
from .x3d import _ClipStemFn, create_x3d, to_logical, to_ndhwc


class _EnhanceFn(torch.autograd.Function):
    """Encoder.enhance (reference model/trainer.py:71-108):
    out = x.clone(); out[:, :, T//2] += relu(conv1x1(|x[:, :, 0] - x[:, :, K+1]|))."""

    @staticmethod
    def forward(ctx, x, weight, t_post):
        ops.require_gpu(x, "enhance input")
        B, C, T, H, W = x.shape
        act = x.dtype
        dt = ops.dt_code(act)
        dev = x.device
        xc = to_ndhwc(x.detach())
        t_mid = T // 2
        M2 = B * H * W
        d = torch.empty((M2, cpad(C)), dtype=act, device=dev)
        ops.frame_absdiff(xc, d, B, T, H * W, cpad(C), 0, t_post, dt)
        e = torch.empty_like(d)
        ops.pw_gemm(d, weight, e, M=M2, K=C, N=C, w_sn=C, w_sk=1, dtype=dt)
        out = torch.empty_like(xc)
        ops.enhance_apply(xc, e, out, B, T, H * W, cpad(C), t_mid, dt)
        ctx.saved = (xc, d, e, weight)
        ctx.meta = (B, C, T, H, W, t_post, t_mid)
        return to_logical(out)

    @staticmethod
    def backward(ctx, dout):
        xc, d, e, weight = ctx.saved
        B, C, T, H, W, t_post, t_mid = ctx.meta
        act = xc.dtype
        dt = ops.dt_code(act)
        doc = to_ndhwc(dout).to(act)
        M2 = B * H * W
        de = torch.empty_like(d)
        ops.enhance_bwd_mask(doc, e, de, B, T, H * W, cpad(C), t_mid, dt)
        dd = torch.empty_like(d)
        ops.pw_gemm(de, weight, dd, M=M2, K=C, N=C, w_sn=1, w_sk=C, dtype=dt)
        gw = ops.grad_of(weight)
        ops.side_run(lambda: ops.pw_wgrad(de, d, gw, M=M2, K=C, N=C, dw_sn=C, dw_sk=1, dtype=dt), de, d)
        dx = torch.empty_like(xc)
        ops.enhance_bwd_apply(doc, xc, dd, dx, B, T, H * W, cpad(C), 0, t_post, dt)
        return to_logical(dx), None, None


def _ndhwc_storage(t):
    """[B,T,H,W,C] contiguous view of a logical NCDHW tensor whose memory already is channels-last
    (None otherwise)."""
    p = t.permute(0, 2, 3, 4, 1)
    return p if p.is_contiguous() else None


# False: _TapFrames never writes into the gradient it receives (set it when hooks / retain_grad observe stage outputs:
# a kept reference to that gradient would otherwise see the tapped-frame terms added later).  Costs one copy per stage.
TAP_INPLACE = True


class _TapFrames(torch.autograd.Function):
    """x -> (x, x[:, :, k0], ..., x[:, :, k0+n-1]) with the frame gradients accumulated by ONE strided
    HIP pass per tapped frame (`c3d_frame_scatter`, accumulate=1) into the gradient that arrives for x.
    The plain `x[:, :, k]` of reference model/trainer.py:136-139 costs a full-size zero fill, a scatter
    and a full-size add per stage in backward (1.3 GB of traffic for the 256x256 stem output).

    Ownership: the gradient arriving for x is the tensor the NEXT stage's backward allocated for exactly
    this edge (x has one other consumer), so it is accumulated into in place; anything else (a gradient
    in a foreign layout / dtype) is first copied into a buffer this function owns.  For the last stage x
    has no other consumer: the gradient buffer is created here (materialize_grads is off)."""

    @staticmethod
    def forward(ctx, x, k0, n):
        ctx.set_materialize_grads(False)
        ctx.k0, ctx.n, ctx.x_meta = k0, n, (tuple(x.shape), x.dtype)
        return (x.view_as(x),) + tuple(x[:, :, k0 + i] for i in range(n))

    @staticmethod
    def backward(ctx, gx, *gf):
        (B, C, T, H, W), act = ctx.x_meta
        live = [(i, g) for i, g in enumerate(gf) if g is not None]
        if gx is None and not live:
            return None, None, None
        dev = (gx if gx is not None else live[0][1]).device
        dt = ops.dt_code(act)
        fresh = gx is None
        # in-place accumulation only into a gradient nobody else can observe: the buffer the next stage's backward just
        # produced for this edge (no autograd history, not a leaf's .grad).  A gradient that went through a tensor hook /
        # retain_grad (requires_grad or grad_fn set under create_graph) or any foreign layout is copied first.
        own = TAP_INPLACE and (not fresh) and gx.dtype == act and not gx.requires_grad and gx.grad_fn is None
        buf = _ndhwc_storage(gx) if own else None
        if buf is None:
            buf = torch.empty((B, T, H, W, C), dtype=act, device=dev)
            if fresh:   # frames nobody tapped carry no gradient
                tapped = {ctx.k0 + i for i, _ in live}
                for t in range(T):
                    if t not in tapped:
                        buf[:, t].zero_()
            else:
                buf.copy_(gx.permute(0, 2, 3, 4, 1))
        for i, g in live:
            gc = g.permute(0, 2, 3, 1)
            if g.dtype != act or not gc.is_contiguous():
                gc = gc.to(act).contiguous()
            ops.frame_scatter(gc, buf, B, T, H * W, C, ctx.k0 + i, 0 if fresh else 1, dt)
        return buf.permute(0, 4, 1, 2, 3), None, None


def tap_frames(x, k0, n):
    out = _TapFrames.apply(x, k0, n)
    return out[0], list(out[1:])


class Encoder(nn.Module):
    """Encoder model based on X3D architecture with feature enhancement capabilities."""

    def __init__(self, args: Any, embed_dims: List[int]) -> None:
        super().__init__()
        self.args = args
        act_dtype = getattr(args, "act_dtype", torch.float32)
        self.x3d = create_x3d(input_clip_length=3, depth_factor=5.0, act_dtype=act_dtype)
        try:  # reference model/trainer.py:43-48 (failure is reported and training continues)
            state_dict = torch.load(args.pretrained, map_location="cpu")["model_state"]
            msg = self.x3d.load_state_dict(state_dict, strict=True)
            print(f"Load pretrained weight: {args.pretrained}, {msg}.")
        except Exception as e:
            print(f"Failed to load pretrained weights: {e}")
        self.perception_frames = nn.Parameter(
            torch.randn(1, 3, args.num_perception_frame, args.in_height, args.in_width), requires_grad=True)
        self.fc = nn.ModuleList([
            nn.Sequential(nn.Conv2d(dim, dim, kernel_size=1, stride=1, padding=0, bias=False), nn.ReLU())
            for dim in embed_dims])
        # the stem only needs the input gradient of the perception frames
        self.x3d.blocks[0].grad_frames = (1, args.num_perception_frame)

    def enhance(self, x: torch.Tensor, fc: nn.Module) -> torch.Tensor:
        return _EnhanceFn.apply(x, fc[0].weight, self.args.num_perception_frame + 1)

    def base_forward(self, x: torch.Tensor, output_final: bool = False, _stem_out=None):
        """`_stem_out`: blocks[0] already applied (the fused clip + stem path of `forward`)."""
        if output_final:
            for i in range(5):
                x = _stem_out if (i == 0 and _stem_out is not None) else self.x3d.blocks[i](x)
            return x[:, :, self.args.num_perception_frame]
        out = []
        for i in range(4):
            x = _stem_out if (i == 0 and _stem_out is not None) else self.x3d.blocks[i](x)
            x = self.enhance(x, self.fc[i])
            x, frames = tap_frames(x, 1, self.args.num_perception_frame)
            out.append(frames)
        return out

    def forward(self, x: torch.Tensor, y: torch.Tensor, output_final: bool = False):
        stem = self.x3d.blocks[0]
        if (x.is_cuda and not x.requires_grad and not y.requires_grad and x.dim() == 4 and x.shape[1] == 3
                and (x.shape[2] * x.shape[3]) % 4 == 0 and tuple(x.shape[2:]) == tuple(self.perception_frames.shape[3:])):
            # torch.cat([x, perception_frames.expand(B), y], dim=2) + blocks[0] as one function: HIP clip assembly,
            # batch-summed perception-frame gradient written by the stem's own backward kernel
            s0 = _ClipStemFn.apply(x, y, self.perception_frames, stem.norm.weight, stem)
            return self.base_forward(None, output_final, _stem_out=s0)
        expand = self.perception_frames.expand(x.shape[0], -1, -1, -1, -1)
        frames = torch.cat([x.unsqueeze(2), expand, y.unsqueeze(2)], dim=2)
        return self.base_forward(frames, output_final)


class Trainer(nn.Module):
    """Complete model with encoder and decoder(s)."""

    def __init__(self, args: Any) -> None:
        super().__init__()
        self.args = args
        self.embed_dims = [24, 24, 48, 96]
        self.encoder = Encoder(args, self.embed_dims)
        k = args.num_perception_frame
        if k == 1 and "CD" in args.dataset:
            self.decoder = ChangeDecoder(args, in_dim=self.embed_dims, has_sigmoid=True)
            weight_init(self.decoder)
        elif k == 3:
            self.decoder_pre = ChangeDecoder(args, in_dim=self.embed_dims)
            self.decoder_post = ChangeDecoder(args, in_dim=self.embed_dims)
            self.decoder_change = ChangeDecoder(args, in_dim=self.embed_dims, has_sigmoid=True)
            weight_init(self.decoder_pre)
            weight_init(self.decoder_post)
            weight_init(self.decoder_change)
        elif k == 2:
            self.decoder_cls = ChangeDecoder(args, in_dim=self.embed_dims)
            self.decoder_loc = ChangeDecoder(args, in_dim=self.embed_dims, has_sigmoid=True)
            weight_init(self.decoder_cls)
            weight_init(self.decoder_loc)
        elif k == 1 and "CC" in args.dataset:   # change captioning (reference model/trainer.py:217-218)
            from .caption_decoder import CaptionDecoder
            self.decoder = CaptionDecoder(args)
        else:
            assert False

    def set_act_dtype(self, dtype):
        self.encoder.x3d.set_act_dtype(dtype)
        return self

    def update_bcd(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        features = self.encoder(x, y)
        return self.decoder([f[0] for f in features])

    def update_scd(self, x: torch.Tensor, y: torch.Tensor):
        features = self.encoder(x, y)
        pre_mask = self.decoder_pre([f[0] for f in features])
        post_mask = self.decoder_post([f[2] for f in features])
        change_mask = self.decoder_change([f[1] for f in features])
        return pre_mask, post_mask, change_mask

    def update_cc(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """reference model/trainer.py:292-306: X3D blocks 0..4 without enhancement, the perception frame of res5
        -> (B, 192, H/16, W/16)."""
        return self.encoder(x, y, output_final=True)

    def update_bda(self, x: torch.Tensor, y: torch.Tensor):
        features = self.encoder(x, y)
        return self.decoder_cls([f[0] for f in features]), self.decoder_loc([f[1] for f in features])
