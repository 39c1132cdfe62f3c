"""MI355X-native mirror of the tensor-side half of the reference's `data/transforms.py` (reference
data/transforms.py:100-154) for the BCD train step: the reference runs random_flip -> random_exchange ->
normalize -> to_tensor per sample on the host (numpy / cv2) inside DataLoader workers; here the raw uint8 batch is
copied to HBM once and ONE HIP pass (`c3d_bcd_preprocess`, csrc/data_ops.hip) produces the normalised float
tensors `Trainer.update_bcd` consumes.  Geometry-changing transforms (scale / resize / random_crop_resize) stay on
the host side of the boundary (they are cv2 interpolation, outside SURVEY.md section 8).

`BCDTransforms.DEFAULT_MEAN/STD` and `IMAGENET_MEAN/STD` are the reference's constants."""
import numpy as np
import torch

from .. import ops


class BCDTransforms:
    DEFAULT_MEAN = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    DEFAULT_STD = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    IMAGENET_MEAN = [0.406, 0.456, 0.485, 0.406, 0.456, 0.485]
    IMAGENET_STD = [0.225, 0.224, 0.229, 0.225, 0.224, 0.229]


def draw_augmentation_flags(batch, rng, train=True):
    """Per-sample (flip0, flip1, exchange) decisions, each with probability 0.5 as in the reference's
    random_flip / random_exchange (data/transforms.py:100-124); all zero for the validation transform."""
    if not train:
        return np.zeros((batch, 3), dtype=np.uint8)
    return (rng.random((batch, 3)) < 0.5).astype(np.uint8)


class DeviceBatchTransform:
    """`(image6 u8 [B,H,W,6], label u8 [B,H,W], flags u8 [B,3]) -> (pre, post, label)` float tensors on the GPU."""

    def __init__(self, device, mean=BCDTransforms.DEFAULT_MEAN, std=BCDTransforms.DEFAULT_STD):
        self.device = torch.device(device)
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device)

    def __call__(self, image6, label=None, flags=None):
        image6 = torch.as_tensor(image6).to(self.device, non_blocking=True).contiguous()
        ops.require_gpu(image6, "raw image batch")
        if image6.dtype != torch.uint8 or image6.dim() != 4 or image6.shape[-1] != 6:
            raise ValueError("image6 must be uint8 [B, H, W, 6] (pre RGB | post RGB)")
        B, H, W, _ = image6.shape
        if label is not None:
            label = torch.as_tensor(label).to(self.device, non_blocking=True).contiguous()
            if label.dtype != torch.uint8 or tuple(label.shape) != (B, H, W):
                raise ValueError("label must be uint8 [B, H, W]")
        if flags is not None:
            flags = torch.as_tensor(flags).to(self.device, non_blocking=True).contiguous()
            if flags.dtype != torch.uint8 or tuple(flags.shape) != (B, 3):
                raise ValueError("flags must be uint8 [B, 3]")
        pre = torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device)
        post = torch.empty_like(pre)
        lab = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device) if label is not None else None
        ops.bcd_preprocess(image6, label, flags, self.mean, self.std, pre, post, lab, B, H, W)
        return pre, post, lab


class SCDTransforms(BCDTransforms):
    """reference data/transforms.py:210-224: same normalisation constants as BCDTransforms."""


class DeviceSCDBatchTransform:
    """`(image6 u8 [B,H,W,6], label3 u8 [B,H,W,3], flags u8 [B,3]) -> (pre, post f32 [B,3,H,W], labels int64 [B,3,H,W])`
    on the GPU: the tensor side of reference SCDTransforms (data/transforms.py:300-357: random_flip, random_exchange --
    which also swaps the two class maps --, normalize, to_tensor) and the `.long()` of scripts/train_SCD.py:207-213.
    The image runs through the BCD pass (identical arithmetic), the labels through `c3d_scd_label_preprocess`."""

    def __init__(self, device, mean=SCDTransforms.DEFAULT_MEAN, std=SCDTransforms.DEFAULT_STD):
        self.image = DeviceBatchTransform(device, mean, std)
        self.device = self.image.device

    def __call__(self, image6, label3, flags=None):
        label3 = torch.as_tensor(label3).to(self.device, non_blocking=True).contiguous()
        B, H, W = label3.shape[:3]
        if label3.dtype != torch.uint8 or label3.dim() != 4 or label3.shape[-1] != 3:
            raise ValueError("label3 must be uint8 [B, H, W, 3] (pre classes, post classes, change)")
        if flags is not None:
            flags = torch.as_tensor(flags).to(self.device, non_blocking=True).contiguous()
        pre, post, _ = self.image(image6, None, flags)
        labels = torch.empty((B, 3, H, W), dtype=torch.int64, device=self.device)
        ops.scd_label_preprocess(label3, flags, labels, B, H, W)
        return pre, post, labels


def cc_normalize_table(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """f32 [3, 256]: Normalize(mean, std)(FloatTensor(u8 / 255.)) for every byte value, with the reference's own
    arithmetic (data/dataset.py:413: numpy u8 / 255. is a float64 division, rounded to f32 by FloatTensor;
    scripts/train_CC.py:466-469 / torchvision Normalize: f32 `sub_(mean).div_(std)`)."""
    v = torch.from_numpy(np.arange(256, dtype=np.uint8) / 255.).to(torch.float32)          # f64 divide -> f32
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1)
    return (v.view(1, 256) - m) / s


class DeviceCCBatchTransform:
    """`(img u8 [B,2,3,H,W], swap u8 [B] or None) -> (pre, post)` f32 [B,3,H,W] on the GPU: reference
    data/dataset.py:411-424 (`/ 255.`, per-image Normalize, the TRAIN split's pair swap) in one HIP pass."""

    def __init__(self, device, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.device = torch.device(device)
        self.lut = cc_normalize_table(mean, std).to(self.device).contiguous()

    def __call__(self, img, swap=None):
        img = torch.as_tensor(img).to(self.device, non_blocking=True).contiguous()
        ops.require_gpu(img, "raw image pair batch")
        if img.dtype != torch.uint8 or img.dim() != 5 or tuple(img.shape[1:3]) != (2, 3):
            raise ValueError("img must be uint8 [B, 2, 3, H, W]")
        B, _, _, H, W = img.shape
        if swap is not None:
            swap = torch.as_tensor(swap).to(self.device, non_blocking=True).contiguous()
            if swap.dtype != torch.uint8 or tuple(swap.shape) != (B,):
                raise ValueError("swap must be uint8 [B]")
        pre = torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device)
        post = torch.empty_like(pre)
        ops.cc_preprocess(img, swap, self.lut, pre, post, B, H, W)
        return pre, post
