#!/usr/bin/env python
"""One bf16 BCD train step on seeded synthetic data; saves loss, every gradient and every buffer to argv[1] (.pt).
Used by tests/test_model_gpu.py to compare library modes (environment knobs are read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib, io
import torch
from change3d_amd import synthetic as synth
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import BCEDiceLoss

size, batch = int(sys.argv[2]), int(sys.argv[3])
args = synth.make_args(size=size, act_dtype=torch.bfloat16)
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    net = Trainer(args)
net.load_state_dict(synth.synth_state_dict(net, seed=5, mask_margin=0.25, branch_gain=0.1))
net = net.to("cuda:0").train()
pre, post, tgt = (t.to("cuda:0") for t in synth.synth_batch(batch, size, seed=2))
loss = BCEDiceLoss(net.update_bcd(pre, post), tgt)
loss.backward()
torch.cuda.synchronize()
torch.save({"loss": loss.detach().cpu(), "grads": {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None},
            "bufs": {n: b.cpu() for n, b in net.named_buffers()}}, sys.argv[1])
