#!/usr/bin/env python
"""Eval-mode (inference) throughput of the BCD path (reference scripts/train_BCD.py:92-154
`val`): model.eval(), torch.no_grad(), update_bcd + binarise + on-device confusion matrix.
usage: python tools/infer_bench.py [batch=32] [steps=20] [dtype=bf16|f32]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from change3d_amd.model.trainer import Trainer  # noqa: E402
from change3d_amd.utils.metric_tool import ConfuseMatrixMeter  # noqa: E402
from change3d_amd import synthetic as synth  # deterministic synthetic weights / batches (neutral module)
from change3d_amd.synthetic import make_args


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
    args = make_args(size=256)
    args.act_dtype = dt
    net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
    net = net.cuda().eval()
    pre, post, tgt = (t.cuda() for t in synth.synth_batch(batch, 256, seed=0))
    meter = ConfuseMatrixMeter(2)
    with torch.no_grad():
        for _ in range(3):
            meter.update_cm_device(net.update_bcd(pre, post), tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            meter.update_cm_device(net.update_bcd(pre, post), tgt)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
    print(json.dumps({"metric": "inference images/sec (256x256 pairs, X3D-L BCD, eval mode)", "value": round(batch * steps / dt_s, 1),
                      "ms_per_batch": round(dt_s / steps * 1e3, 3), "ms_per_sample": round(dt_s / steps / batch * 1e3, 4),
                      "batch": batch, "dtype": str(dt).split(".")[-1]}))


if __name__ == "__main__":
    main()
