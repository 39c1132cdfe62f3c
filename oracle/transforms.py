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
