import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model as om, synth
from change3d_amd import ops
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import ChangeSimilarity, CrossEntropyLoss2d
from change3d_amd.scripts.train_SCD import scd_loss
DEV = "cuda:0"
size, batch = 64, 2
mk = lambda: om.make_args(num_perception_frame=3, size=size, dataset="SECOND", num_class=7)
ref = om.Trainer(mk())
sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25, branch_gain=0.1)
ref.load_state_dict(sd); ref.train()
pre, post, tgt = synth.synth_batch(batch, size, seed=0)
labels = synth.synth_scd_labels(batch, size, seed=0)
om.scd_loss(*ref.update_scd(pre, post), labels).backward()
pref = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
res = {}
for m in ("1", "0"):
    ops.set_option(ops.OPT_STEM_MFMA, int(m))
    mine = Trainer(mk()); mine.load_state_dict(sd); mine = mine.to(DEV).train()
    o_d = mine.update_scd(pre.to(DEV), post.to(DEV))
    scd_loss(CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity(), o_d, labels.to(DEV))[0].backward()
    torch.cuda.synchronize()
    res[m] = {n: p.grad.detach().cpu().clone() for n, p in mine.named_parameters() if p.grad is not None}
    res[m + "buf"] = {n: b.detach().cpu().clone().double() for n, b in mine.named_buffers()}
    res[m + "out"] = [o.detach().cpu().clone() for o in o_d]
for n in ("encoder.x3d.blocks.2.res_blocks.6.branch2.norm_a.bias", "encoder.x3d.blocks.2.res_blocks.6.branch2.norm_a.weight",
          "encoder.x3d.blocks.1.res_blocks.0.branch2.norm_b.1.block.0.bias"):
    r = pref[n].flatten()
    for m in ("1", "0"):
        d = (res[m][n].flatten() - r).abs()
        top = torch.topk(d, min(4, d.numel()))
        print(n.split("blocks.", 1)[1], "MFMA=" + m, "norm", f"{r.norm():.3e}", "top |diff|", [f"{v:.2e}@{i}" for v, i in zip(top.values.tolist(), top.indices.tolist())],
              "rest", f"{(d.pow(2).sum() - top.values[0] ** 2).clamp(min=0).sqrt():.2e}")

bd = {n: (res["1buf"][n] - res["0buf"][n]).abs().max().item() / (res["0buf"][n].abs().max().item() + 1e-30) for n in res["1buf"]}
top = sorted(bd.items(), key=lambda kv: -kv[1])[:6]
print("buffers MFMA vs scalar (rel max diff):", [(n[-60:], f"{v:.1e}") for n, v in top])
print("buffers identical:", sum(1 for v in bd.values() if v == 0), "of", len(bd))
print("outputs MFMA vs scalar:", [f"{(a - b).abs().max().item():.1e}" for a, b in zip(res["1out"], res["0out"])])
gd = {n: ((res["1"][n] - res["0"][n]).norm() / (res["0"][n].norm() + 1e-30)).item() for n in res["1"]}
print("grads identical:", sum(1 for v in gd.values() if v == 0), "of", len(gd), "worst", sorted(gd.items(), key=lambda kv: -kv[1])[:3])
