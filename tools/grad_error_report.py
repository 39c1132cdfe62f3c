import sys, os, torch
sys.path.insert(0, os.getcwd())
from oracle import model as om, synth
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import BCEDiceLoss, hot_path_named_params
args = om.make_args(size=64)
ref = om.Trainer(args); sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25); ref.load_state_dict(sd); ref.train()
pre, post, tgt = synth.synth_batch(2, 64, seed=0)
om.bce_dice_loss(ref.update_bcd(pre, post), tgt).backward()
r64 = om.Trainer(args); r64.load_state_dict(sd); r64 = r64.double().train()
om.bce_dice_loss(r64.update_bcd(pre.double(), post.double()), tgt.double()).backward()
net = Trainer(args); net.load_state_dict(sd); net = net.cuda().train()
BCEDiceLoss(net.update_bcd(pre.cuda(), post.cuda()), tgt.cuda()).backward()
g32, g64 = dict(ref.named_parameters()), dict(r64.named_parameters())
rows = []
for n, p in hot_path_named_params(net):
    a = p.grad.double().cpu(); b = g64[n].grad; c = g32[n].grad.double()
    rows.append(((a-b).norm().item()/(b.norm().item()+1e-30), (c-b).norm().item()/(b.norm().item()+1e-30), b.norm().item(), n))
rows.sort(reverse=True)
for r in rows[:14]: print(f"hip {r[0]:.2e}  ref32 {r[1]:.2e}  |g64| {r[2]:.3e}  {r[3]}")
import collections
kinds = collections.defaultdict(list)
for r in rows:
    k = r[3].split('.')[-2] + '.' + r[3].split('.')[-1] if 'norm' in r[3] else r[3].split('.')[-2]
    kinds[k].append(r[0] / max(r[1], 1e-12))
for k, v in sorted(kinds.items(), key=lambda kv: -sum(kv[1])/len(kv[1]))[:12]:
    v = sorted(v); print(f"{k:28s} n={len(v):3d} median hip/ref ratio {v[len(v)//2]:.2f}  max {v[-1]:.2f}")
