#!/usr/bin/env python
"""Run-to-run reproducibility probe (GPU box): repeats one forward+backward from identical state and
reports, per parameter, how far the gradients of later runs move from the first run.  f32 atomics make
tiny differences legitimate (1e-6 relative); anything larger is a race.
usage: python tools/determinism.py [size=64] [batch=2] [runs=12] [dtype=f32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from change3d_amd.model.trainer import Trainer  # noqa: E402
from change3d_amd.model.utils import BCEDiceLoss  # noqa: E402


class Args:
    pass


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
    a = Args()
    a.pretrained, a.in_height, a.in_width, a.num_class, a.num_perception_frame = "/nonexistent", size, size, 1, 1
    a.dataset = "LEVIR-CD"
    a.act_dtype = dt
    torch.manual_seed(0)
    m = Trainer(a).to("cuda:0")
    m.train()
    g = torch.Generator(device="cpu").manual_seed(1)
    pre = torch.randn(batch, 3, size, size, generator=g).cuda()
    post = torch.randn(batch, 3, size, size, generator=g).cuda()
    tgt = (torch.rand(batch, 1, size, size, generator=g) > 0.5).float().cuda()
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    ref = None
    worst = {}
    state = {k: v.clone() for k, v in m.state_dict().items()}   # BN running buffers move every forward
    if os.environ.get("C3D_TRACE"):
        from change3d_amd import ops
        traces = []
        for r in range(runs):
            m.load_state_dict(state)
            for p in m.parameters():
                p.grad = None
            ops.trace_begin()
            loss = BCEDiceLoss(m.update_bcd(pre, post), tgt)
            loss.backward()
            traces.append(ops.trace_end())
        t0 = traces[0]
        for r, t in enumerate(traces[1:], 1):
            assert len(t) == len(t0)
            for k, ((n0, c0), (n1, c1)) in enumerate(zip(t0, t)):
                thr = float(os.environ.get("C3D_TRACE_TOL", "0"))
                if any(abs(a_[2] - b_[2]) > thr * max(abs(a_[2]), 1e-30) for a_, b_ in zip(c0, c1)):
                    print(f"run {r}: first difference (> {thr:g} rel) at launch {k}/{len(t0)} {n0}")
                    for a_, b_ in zip(c0, c1):
                        print(f"     {a_[0]} {a_[1]} {a_[2]!r} vs {b_[2]!r} {'<--' if a_[2] != b_[2] else ''}")
                    for kk in range(max(0, k - 3), k):
                        print(f"   before: {kk} {t0[kk][0]}")
                    break
            else:
                print(f"run {r}: identical trace ({len(t0)} launches)")
        return
    for r in range(runs):
        m.load_state_dict(state)
        for p in m.parameters():
            p.grad = None
        out = m.update_bcd(pre, post)
        loss = BCEDiceLoss(out, tgt)
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
        if ref is None:
            ref = grads
            lref = loss.item()
            continue
        es = []
        for n, gd in grads.items():
            e = (gd - ref[n]).norm().item() / (ref[n].norm().item() + 1e-30)
            worst[n] = max(worst.get(n, 0.0), e)
            es.append(e)
        es.sort()
        print(f"run {r}: loss diff {loss.item() - lref:+.3e}  grad rel dev vs run 0: median {es[len(es) // 2]:.2e} "
              f"p90 {es[int(len(es) * 0.9)]:.2e} max {es[-1]:.2e}", flush=True)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:15]
    for n, e in top:
        print(f"  {e:.3e}  {n}")
    nbad = sum(1 for e in worst.values() if e > 1e-4)
    print(f"{nbad}/{len(worst)} parameters moved by more than 1e-4 relative")


if __name__ == "__main__":
    main()
