"""ORACLE -- test infrastructure only.

CPU restatement (numpy) of the tensor-side transforms of the reference's BCD input pipeline, per sample exactly as
reference data/transforms.py:100-154 composes them (scripts/train_BCD.py:262-270: random_flip, random_exchange,
normalize, to_tensor); the random decisions are passed in as flags so both sides see the same draw.  cv2.flip(a, 0)
reverses the rows, cv2.flip(a, 1) the columns (cv2 itself is not installed here: flips are restated as slicing).
Parity unpinned by the reference (it has no tests); pinned by arithmetic identity with numpy."""
import numpy as np


def bcd_transform_sample(image6, label, flags, mean, std):
    """image6 u8 [H,W,6], label u8 [H,W], flags (flip0, flip1, exchange) -> (image f32 [6,H,W], label f32 [1,H,W])."""
    if flags[0]:                      # random_flip: cv2.flip(image, 0)
        image6, label = image6[::-1], label[::-1]
    if flags[1]:                      # cv2.flip(image, 1)
        image6, label = image6[:, ::-1], label[:, ::-1]
    if flags[2]:                      # random_exchange
        image6 = np.concatenate((image6[:, :, 3:6], image6[:, :, 0:3]), axis=2)
    mean_array = np.array(mean, dtype=np.float32).reshape(1, 1, -1)
    std_array = np.array(std, dtype=np.float32).reshape(1, 1, -1)
    image = image6.astype(np.float32) / 255.0          # normalize (reference data/transforms.py:127-137)
    lab = np.ceil(label / 255.0).astype(np.float32)
    image = (image - mean_array) / std_array
    return np.ascontiguousarray(image.transpose((2, 0, 1))), lab[None]   # to_tensor


def bcd_transform_batch(image6, label, flags, mean, std):
    outs = [bcd_transform_sample(image6[b], label[b], flags[b], mean, std) for b in range(image6.shape[0])]
    img = np.stack([o[0] for o in outs])
    return img[:, 0:3], img[:, 3:6], np.stack([o[1] for o in outs])


def scd_transform_sample(image6, label3, flags, mean, std):
    """reference data/transforms.py:300-357 (SCDTransforms: normalize -> random_flip -> random_exchange -> to_tensor;
    the normalisation is per channel with identical pre / post constants, so it commutes with the flips and the
    exchange).  image6 u8 [H,W,6], label3 u8 [H,W,3] -> (image f32 [6,H,W], label int64 [3,H,W] as
    scripts/train_SCD.py:207-213 `.long()` leaves it)."""
    mean_array = np.array(mean, dtype=np.float32).reshape(1, 1, -1)
    std_array = np.array(std, dtype=np.float32).reshape(1, 1, -1)
    image = image6.astype(np.float32) / 255.0
    image = (image - mean_array) / std_array
    if flags[0]:
        image, label3 = image[::-1], label3[::-1]
    if flags[1]:
        image, label3 = image[:, ::-1], label3[:, ::-1]
    if flags[2]:
        image = np.concatenate((image[:, :, 3:6], image[:, :, 0:3]), axis=2)
        label3 = np.concatenate((label3[:, :, 1:2], label3[:, :, 0:1], label3[:, :, 2:3]), axis=2)
    return np.ascontiguousarray(image.transpose((2, 0, 1))), np.ascontiguousarray(label3.transpose((2, 0, 1))).astype(np.int64)


def cc_transform_sample(img, swap, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """reference data/dataset.py:411-424 + scripts/train_CC.py:466-469: img u8 [2,3,H,W] -> f32 [2,3,H,W]:
    torch.FloatTensor(img / 255.) (float64 division, then f32), torchvision Normalize per image (f32 sub, div),
    then the pair swap of the TRAIN split."""
    import torch
    t = torch.FloatTensor(img / 255.)
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    t[0] = (t[0] - m) / s
    t[1] = (t[1] - m) / s
    if swap:
        t[0], t[1] = t[1].clone(), t[0].clone()
    return t.numpy()
