"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the third-party classes the reference's ``model/x3d.py`` imports
(reference ``model/x3d.py:13-20``) but that are NOT vendored under /root/reference and
NOT installed in this image:

  * ``pytorchvideo==0.1.5``  (reference ``requirements.txt:10``)
  * ``fvcore==0.1.5.post20221221`` (reference ``requirements.txt:3``)

They are thin ``torch.nn`` wrappers, restated here from the published behaviour of
those pinned releases.  PARITY UNPINNED at this boundary: the reference ships no tests
or golden vectors for these classes, so the restatement is anchored only on (a) the
reference's own call sites (constructor keywords, ``model/x3d.py:87-92,101-106,
195-199,223-232,300-328,412,534-540,744``), (b) the state-dict key schema those call
sites imply, and (c) the published parameter counts (X3D-L 6.15 M, BCD 1.54 M — see
``tests/test_oracle_cpu.py``).

All arithmetic runs in torch-CPU fp32 eager, logical NCDHW, exactly like the
reference's own CPU path.
"""
import math

import torch
import torch.nn as nn


# ----------------------------------------------------------------------------------
# pytorchvideo.layers.utils
# ----------------------------------------------------------------------------------
def set_attributes(self, params=None):
    """pytorchvideo.layers.utils.set_attributes: copy ctor locals to attributes
    (this is what defines the reference's state-dict key names)."""
    if params:
        for k, v in params.items():
            if k != "self":
                setattr(self, k, v)


def round_width(width, multiplier, min_width=8, divisor=8, ceil=False):
    """pytorchvideo.layers.utils.round_width (used at reference model/x3d.py:197,657,675-683)."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    if ceil:
        width_out = max(min_width, int(math.ceil(width / divisor)) * divisor)
    else:
        width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


def round_repeats(repeats, multiplier):
    """pytorchvideo.layers.utils.round_repeats (reference model/x3d.py:685)."""
    if not multiplier:
        return repeats
    return int(math.ceil(multiplier * repeats))


# ----------------------------------------------------------------------------------
# pytorchvideo.layers.swish
# ----------------------------------------------------------------------------------
class Swish(nn.Module):
    """x * sigmoid(x).  Upstream uses a memory-saving custom autograd function whose
    backward is g * sig(x) * (1 + x * (1 - sig(x))) — the same derivative autograd
    produces for this expression."""

    def forward(self, x):
        return x * torch.sigmoid(x)


# ----------------------------------------------------------------------------------
# fvcore.nn.squeeze_excitation
# ----------------------------------------------------------------------------------
class SqueezeExcitation(nn.Module):
    """fvcore SqueezeExcitation as called at reference model/x3d.py:195-199
    (num_channels, num_channels_reduced, is_3d=True): submodule ``block`` =
    Sequential(conv(C->Cr, bias), ReLU, conv(Cr->C, bias), Sigmoid) applied to the
    global mean; output = input * gate."""

    def __init__(self, num_channels, num_channels_reduced=None, reduction_ratio=2.0,
                 is_3d=False, activation=None):
        super().__init__()
        if num_channels_reduced is None:
            num_channels_reduced = int(num_channels // reduction_ratio)
        if activation is None:
            activation = nn.ReLU()
        if is_3d:
            conv1 = nn.Conv3d(num_channels, num_channels_reduced, kernel_size=1, bias=True)
            conv2 = nn.Conv3d(num_channels_reduced, num_channels, kernel_size=1, bias=True)
        else:
            conv1 = nn.Conv2d(num_channels, num_channels_reduced, kernel_size=1, bias=True)
            conv2 = nn.Conv2d(num_channels_reduced, num_channels, kernel_size=1, bias=True)
        self.is_3d = is_3d
        self.block = nn.Sequential(conv1, activation, conv2, nn.Sigmoid())

    def forward(self, input_tensor):
        dims = [2, 3, 4] if self.is_3d else [2, 3]
        mean_tensor = input_tensor.mean(dim=dims, keepdim=True)
        return torch.mul(input_tensor, self.block(mean_tensor))


# ----------------------------------------------------------------------------------
# pytorchvideo.layers.convolutions
# ----------------------------------------------------------------------------------
class Conv2plus1d(nn.Module):
    """conv_t -> [norm] -> [activation] -> conv_xy (conv_xy_first=False).  NOTE the
    reference stores the *spatial* 1x3x3 conv as ``conv_t`` and the *temporal* 5x1x1
    depthwise conv as ``conv_xy`` (reference model/x3d.py:87-92)."""

    def __init__(self, *, conv_t=None, norm=None, activation=None, conv_xy=None,
                 conv_xy_first=False):
        super().__init__()
        set_attributes(self, locals())
        assert self.conv_t is not None
        assert self.conv_xy is not None

    def forward(self, x):
        x = self.conv_xy(x) if self.conv_xy_first else self.conv_t(x)
        x = self.norm(x) if self.norm else x
        x = self.activation(x) if self.activation else x
        x = self.conv_t(x) if self.conv_xy_first else self.conv_xy(x)
        return x


# ----------------------------------------------------------------------------------
# pytorchvideo.models.stem / resnet / head / net
# ----------------------------------------------------------------------------------
class ResNetBasicStem(nn.Module):
    """conv -> norm -> activation -> pool (reference model/x3d.py:101-106)."""

    def __init__(self, *, conv=None, norm=None, activation=None, pool=None):
        super().__init__()
        set_attributes(self, locals())
        assert self.conv is not None

    def forward(self, x):
        x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        if self.pool is not None:
            x = self.pool(x)
        return x


class BottleneckBlock(nn.Module):
    """conv_a,norm_a,act_a, conv_b,norm_b,act_b, conv_c,norm_c (reference model/x3d.py:223-232)."""

    def __init__(self, *, conv_a=None, norm_a=None, act_a=None, conv_b=None, norm_b=None,
                 act_b=None, conv_c=None, norm_c=None):
        super().__init__()
        set_attributes(self, locals())
        assert all(op is not None for op in (self.conv_a, self.conv_b, self.conv_c))

    def forward(self, x):
        x = self.conv_a(x)
        if self.norm_a is not None:
            x = self.norm_a(x)
        if self.act_a is not None:
            x = self.act_a(x)
        x = self.conv_b(x)
        if self.norm_b is not None:
            x = self.norm_b(x)
        if self.act_b is not None:
            x = self.act_b(x)
        x = self.conv_c(x)
        if self.norm_c is not None:
            x = self.norm_c(x)
        return x


class ResBlock(nn.Module):
    """fusion(shortcut, branch2(x)) -> activation (reference model/x3d.py:300-328)."""

    def __init__(self, branch1_conv=None, branch1_norm=None, branch2=None, activation=None,
                 branch_fusion=None):
        super().__init__()
        set_attributes(self, locals())
        assert self.branch2 is not None

    def forward(self, x):
        if self.branch1_conv is None:
            x = self.branch_fusion(x, self.branch2(x))
        else:
            shortcut = self.branch1_conv(x)
            if self.branch1_norm is not None:
                shortcut = self.branch1_norm(shortcut)
            x = self.branch_fusion(shortcut, self.branch2(x))
        if self.activation is not None:
            x = self.activation(x)
        return x


class ResStage(nn.Module):
    def __init__(self, res_blocks):
        super().__init__()
        self.res_blocks = res_blocks

    def forward(self, x):
        for res_block in self.res_blocks:
            x = res_block(x)
        return x


class ResNetBasicHead(nn.Module):
    """pool -> dropout -> proj (channels-last) -> activation -> output_pool.
    Constructed and strict-loaded by the reference, never executed by any Change3D
    task (reference model/trainer.py:128 iterates range(4); CC uses range(5))."""

    def __init__(self, pool=None, dropout=None, proj=None, activation=None, output_pool=None):
        super().__init__()
        set_attributes(self, locals())
        assert self.proj is not None

    def forward(self, x):
        if self.pool is not None:
            x = self.pool(x)
        if self.dropout is not None:
            x = self.dropout(x)
        if self.proj is not None:
            x = x.permute((0, 2, 3, 4, 1))
            x = self.proj(x)
            x = x.permute((0, 4, 1, 2, 3))
        if self.activation is not None:
            x = self.activation(x)
        if self.output_pool is not None:
            x = self.output_pool(x)
            x = x.view(x.shape[0], -1)
        return x


def _init_resnet_weights(model, fc_init_std=0.01):
    """pytorchvideo.layers.utils / models.weight_init restatement: Conv3d ->
    kaiming-normal fan_out (c2_msra_fill), norm -> weight 1 / bias 0, Linear ->
    N(0, fc_init_std).  Only matters for random-init construction; every parity
    fixture overwrites all weights from oracle/synth.py."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            if m.weight is not None:
                m.weight.data.fill_(1.0)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=fc_init_std)
            if m.bias is not None:
                m.bias.data.zero_()
    return model


class Net(nn.Module):
    def __init__(self, *, blocks):
        super().__init__()
        assert blocks is not None
        self.blocks = blocks
        _init_resnet_weights(self)

    def forward(self, x):
        for block in self.blocks:
            x = block(x)
        return x
