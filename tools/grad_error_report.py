"""Per-parameter gradient error of the f32 HIP path against the fp64 oracle, with the information needed to tell
CONDITIONING from a KERNEL defect (round-1 review, weak item 2: the worst parameter sat at 6.3e-2 rel-L2 against
1.8e-2 for the fp32 CPU oracle):

  hip      |g_hip  - g_fp64| / |g_fp64|      the HIP f32 path
  ref32    |g_ref32 - g_fp64| / |g_fp64|     the fp32 CPU oracle (same arithmetic precision, ATen summation order)
  cond     |g_fp64(perturbed) - g_fp64| / |g_fp64| for a relative perturbation of 2^-24 (half an f32 ulp, i.i.d.) of
           every parameter and input in the fp64 oracle: what ONE rounding of the inputs alone does to this gradient.
           A parameter whose `cond` is of the order of its `hip` error is ill-conditioned: no f32 implementation can
           do better than a small multiple of it, whatever its kernels do.
  hip/cond, ref32/cond   errors in units of that floor: a kernel defect shows as hip/cond >> ref32/cond for ONE
           kernel family (all conv_b weights, say); conditioning shows as both ratios of order 1-10 everywhere.

Run on the GPU box:  python tools/grad_error_report.py [size] > gpurun_out/grad_error_report.txt
(oracle/ is used here as the checker only; this is a diagnosis tool, not product code.)"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from oracle import model as om  # noqa: E402
from change3d_amd import synthetic as synth  # noqa: E402
from change3d_amd.model.trainer import Trainer  # noqa: E402
from change3d_amd.model.utils import BCEDiceLoss, hot_path_named_params  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
args = om.make_args(size=size)
ref = om.Trainer(args)
sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25)
ref.load_state_dict(sd)
ref.train()
pre, post, tgt = synth.synth_batch(2, size, seed=0)
om.bce_dice_loss(ref.update_bcd(pre, post), tgt).backward()


def fp64_grads(noise_seed=None):
    net = om.Trainer(args)
    net.load_state_dict(sd)
    net = net.double().train()
    a, b = pre.double(), post.double()
    if noise_seed is not None:
        g = torch.Generator().manual_seed(noise_seed)
        eps = 2.0 ** -24
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.0 + eps * torch.randn(p.shape, generator=g, dtype=torch.float64))
        a = a * (1.0 + eps * torch.randn(a.shape, generator=g, dtype=torch.float64))
        b = b * (1.0 + eps * torch.randn(b.shape, generator=g, dtype=torch.float64))
    om.bce_dice_loss(net.update_bcd(a, b), tgt.double()).backward()
    return {n: p.grad for n, p in net.named_parameters() if p.grad is not None}


g64 = fp64_grads()
pert = [fp64_grads(s) for s in (1, 2)]
net = Trainer(args)
net.load_state_dict(sd)
net = net.cuda().train()
BCEDiceLoss(net.update_bcd(pre.cuda(), post.cuda()), tgt.cuda()).backward()
torch.cuda.synchronize()
g32 = dict(ref.named_parameters())


def rl2(a, b):
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


rows = []
for n, p in hot_path_named_params(net):
    b = g64[n]
    cond = max(rl2(q[n], b) for q in pert)
    rows.append(dict(name=n, hip=rl2(p.grad.double().cpu(), b), ref=rl2(g32[n].grad.double(), b), cond=cond,
                     norm=b.norm().item()))
rows.sort(key=lambda r: -r["hip"])
print(f"# gradient error report, BCD {size}x{size} B=2, f32 HIP path vs fp64 oracle ({len(rows)} parameters)")
hip = np.array([r["hip"] for r in rows]); rf = np.array([r["ref"] for r in rows]); cd = np.array([r["cond"] for r in rows])
print(f"# median: hip {np.median(hip):.2e}  ref32 {np.median(rf):.2e}  cond(2^-24 input rounding) {np.median(cd):.2e}")
print(f"# max:    hip {hip.max():.2e}  ref32 {rf.max():.2e}  cond {cd.max():.2e}")
print(f"# median hip/cond {np.median(hip / cd):.1f}   median ref32/cond {np.median(rf / cd):.1f}   "
      f"corr(log hip, log cond) {np.corrcoef(np.log(hip), np.log(cd))[0, 1]:.3f}   "
      f"corr(log ref32, log cond) {np.corrcoef(np.log(rf), np.log(cd))[0, 1]:.3f}")
print("\n## 20 worst parameters by HIP error")
print(f"{'hip':>9s} {'ref32':>9s} {'cond':>9s} {'hip/cond':>8s} {'ref/cond':>8s} {'|g64|':>10s}  parameter")
for r in rows[:20]:
    print(f"{r['hip']:9.2e} {r['ref']:9.2e} {r['cond']:9.2e} {r['hip'] / r['cond']:8.1f} {r['ref'] / r['cond']:8.1f} "
          f"{r['norm']:10.3e}  {r['name']}")
worst = rows[0]["name"]
prefix = worst.rsplit(".branch", 1)[0] if ".branch" in worst else worst.rsplit(".", 2)[0]
print(f"\n## every parameter of the block holding the worst one ({prefix})")
for r in sorted((r for r in rows if r["name"].startswith(prefix + ".")), key=lambda r: r["name"]):
    print(f"{r['hip']:9.2e} {r['ref']:9.2e} {r['cond']:9.2e} {r['hip'] / r['cond']:8.1f} {r['ref'] / r['cond']:8.1f} "
          f"{r['norm']:10.3e}  {r['name']}")
print("\n## by kernel family (which HIP kernel produces the gradient): error in units of the conditioning floor")
fam = collections.defaultdict(list)
for r in rows:
    n = r["name"]
    if "norm" in n and ("weight" in n or "bias" in n) and "block" not in n.split("norm")[-1]:
        k = "BN " + n.split(".")[-2].replace("branch1_norm", "norm_1") + "." + n.split(".")[-1]
    elif ".norm_b.1.block." in n:
        k = "SE fc " + n.split(".")[-2] + "." + n.split(".")[-1]
    elif "conv_b" in n:
        k = "depthwise conv_b (c3d_dw333_bwd_fused)"
    elif "conv_a" in n or "conv_c" in n or "branch1_conv" in n or ".fc." in n:
        k = "pointwise " + n.split(".")[-2] + " (c3d_pw_wgrad)"
    elif "blocks.0" in n:
        k = "stem " + n.split(".")[-2]
    elif "decoder" in n:
        k = "decoder " + ".".join(n.split(".")[-3:])
    else:
        k = n
    fam[k].append(r)
print(f"{'n':>4s} {'med hip/cond':>12s} {'max hip/cond':>12s} {'med ref/cond':>12s} {'max ref/cond':>12s}  family")
for k, v in sorted(fam.items(), key=lambda kv: -np.median([r['hip'] / r['cond'] for r in kv[1]])):
    hc = np.array([r["hip"] / r["cond"] for r in v]); rc = np.array([r["ref"] / r["cond"] for r in v])
    print(f"{len(v):4d} {np.median(hc):12.1f} {hc.max():12.1f} {np.median(rc):12.1f} {rc.max():12.1f}  {k}")
