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
