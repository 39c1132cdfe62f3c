"""Deterministic synthetic weights / inputs / targets for benchmarks, scripts, smoke and tests.

Neither product compute nor oracle: a data generator both sides share, so that the HIP path and the
CPU oracle see bit-identical tensors for a given seed.  Generated with numpy's PCG64
(`np.random.default_rng`): the same seed yields the same tensors in the build container and on the
GPU box (no dependence on torch's RNG stream).

Input contract mirrors the reference data pipeline (reference data/transforms.py:127-154,
LEVIR normalisation mean=std=0.5): image = (u8/255 - 0.5)/0.5 with u8 ~ U{0..255};
target in {0,1} made of 1-3 axis-aligned rectangles (~5 % of pixels).
`X3D_L.pyth` is not available offline, so weights are synthetic per key.
"""
from types import SimpleNamespace

import numpy as np
import torch


def make_args(num_perception_frame=1, size=256, dataset="LEVIR-CD", num_class=1, **extra):
    """The argparse fields `Trainer(args)` reads (reference model/trainer.py:28-56,175-220); `extra` carries the
    caption-decoder fields of the CC task (vocab_size, embed_dim, n_head, n_layer, dropout: reference
    scripts/train_CC.py:423,553-579)."""
    return SimpleNamespace(pretrained="/nonexistent", num_perception_frame=num_perception_frame,
                           in_height=size, in_width=size, dataset=dataset, num_class=num_class, **extra)


def make_cc_args(size=256, vocab_size=501, dropout=0.1):
    """Change-captioning configuration (BASELINE.json configs[4]; reference scripts/train_CC.py:553-579 defaults:
    embed_dim 192 = res5 width, 8 heads, 3 layers, dropout 0.1; vocabulary = len(WORDMAP), synthetic 501)."""
    return make_args(size=size, dataset="LEVIR-CC", vocab_size=vocab_size, embed_dim=192, n_head=8, n_layer=3,
                     dropout=dropout)


def synth_captions(batch, seed=0, vocab_size=501, max_len=52):
    """(caps int64 [B, 52], caplens int64 [B, 1]) shaped like the LEVIR-CC loader's output (reference
    scripts/train_CC.py:105-116): <start>=vocab-2, words in 1..vocab-3, <end>=vocab-1, then <pad>=0."""
    rng = np.random.default_rng(3000 + seed)
    caps = np.zeros((batch, max_len), dtype=np.int64)
    lens = np.zeros((batch, 1), dtype=np.int64)
    for b in range(batch):
        n = int(rng.integers(5, max_len - 2))
        caps[b, 0] = vocab_size - 2
        caps[b, 1:1 + n] = rng.integers(1, vocab_size - 2, size=n)
        caps[b, 1 + n] = vocab_size - 1
        lens[b, 0] = n + 2
    return torch.from_numpy(caps), torch.from_numpy(lens)


def synth_state_dict(model, seed=16, mask_margin=1.0, branch_gain=1.0):
    """Fill every tensor of `model.state_dict()` from a key-ordered PCG64 stream.

    conv / linear weights: N(0, sqrt(2/fan_in)); BN weight U(0.5,1.5), bias N(0,0.1),
    running_mean N(0,0.1), running_var U(0.5,1.5); conv biases N(0,0.1);
    perception_frames N(0,1).  `mask_margin` scales `*.up_c1.0.weight` so sigmoid
    outputs move away from 0.5 (SURVEY.md §7 'mask parity is fragile').  `branch_gain` scales every
    `norm_c.weight` (the last BatchNorm of each residual branch): 1.0 is the default, chaotic network (each of the
    55 blocks adds a branch as large as its shortcut; a 2^-24 input perturbation moves gradients by ~1e-2), small
    values give a trained-network-like, well-conditioned residual stack for tests that need rounding errors to
    stay in the linear regime (bf16 storage).  The random stream is identical for every value."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, ref in model.state_dict().items():
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            val = np.zeros(shape, dtype=np.int64)
        elif key.endswith("running_mean"):
            val = rng.normal(0.0, 0.1, shape)
        elif key.endswith("running_var"):
            val = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("perception_frames"):
            val = rng.normal(0.0, 1.0, shape)
        elif len(shape) == 1:
            is_norm_w = key.endswith("weight")
            val = rng.uniform(0.5, 1.5, shape) if is_norm_w else rng.normal(0.0, 0.1, shape)
            if key.endswith("norm_c.weight"):
                val = val * branch_gain
        else:
            fan_in = int(np.prod(shape[1:]))
            val = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
            if key.endswith("up_c1.0.weight"):
                val = val * mask_margin
        dtype = torch.int64 if key.endswith("num_batches_tracked") else torch.float32
        out[key] = torch.from_numpy(np.ascontiguousarray(val)).to(dtype)
    return out


def synth_batch(batch, size=256, seed=0):
    """(pre, post, target): (B,3,S,S) fp32 x2 in [-1,1], (B,1,S,S) fp32 in {0,1}."""
    rng = np.random.default_rng(1000 + seed)
    u8 = rng.integers(0, 256, size=(2, batch, 3, size, size), dtype=np.uint8)
    imgs = (u8.astype(np.float32) / 255.0 - 0.5) / 0.5
    tgt = np.zeros((batch, 1, size, size), dtype=np.float32)
    for b in range(batch):
        for _ in range(int(rng.integers(1, 4))):
            h = int(rng.integers(size // 16, size // 4))
            w = int(rng.integers(size // 16, size // 4))
            y0 = int(rng.integers(0, size - h))
            x0 = int(rng.integers(0, size - w))
            tgt[b, 0, y0:y0 + h, x0:x0 + w] = 1.0
    return torch.from_numpy(imgs[0]), torch.from_numpy(imgs[1]), torch.from_numpy(tgt)


def synth_scd_labels(batch, size=256, seed=0, num_class=7):
    """SCD labels as `scripts/train_SCD.py:209-217` sees them after the loader: (B,3,S,S) int64 =
    [pre class map in 0..num_class-1, post class map, change mask in {0,1}] (blocky class maps, the
    change mask of `synth_batch` with the same seed)."""
    rng = np.random.default_rng(2000 + seed)
    cell = max(size // 8, 1)
    grid = rng.integers(0, num_class, size=(2, batch, (size + cell - 1) // cell, (size + cell - 1) // cell))
    maps = np.repeat(np.repeat(grid, cell, axis=2), cell, axis=3)[:, :, :size, :size]
    change = synth_batch(batch, size, seed)[2].numpy()[:, 0].astype(np.int64)
    return torch.from_numpy(np.stack([maps[0], maps[1], change], axis=1).astype(np.int64))


def synth_tensor(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))
