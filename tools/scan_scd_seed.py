"""Scan (weight seed, data seed) pairs for the SCD conditioned-weights parity case: worst per-parameter gradient rel-L2
of the f32 HIP path vs the fp32 oracle (a ReLU pre-activation on the kink shows up as ~100 tensors at 1e-4..1e-3)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import contextlib, io
import torch
from oracle import model as om, synth
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import ChangeSimilarity, CrossEntropyLoss2d, hot_path_named_params
from change3d_amd.scripts.train_SCD import scd_loss
torch.set_num_threads(1)   # the oracle's own f32 rounding depends on the thread count: the test pins it to 1
size, batch = 64, 2
mk = lambda: om.make_args(num_perception_frame=3, size=size, dataset="SECOND", num_class=7)
rel = lambda a, b: ((a.detach().double().cpu() - b.detach().double().cpu()).norm() / (b.detach().double().cpu().norm() + 1e-30)).item()
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 40)
for wseed, dseed in [(s, 0) for s in range(lo, hi)]:
    with contextlib.redirect_stdout(io.StringIO()):
        ref, mine = om.Trainer(mk()), Trainer(mk())
    sd = synth.synth_state_dict(ref, seed=wseed, mask_margin=0.25, branch_gain=0.1)
    ref.load_state_dict(sd); mine.load_state_dict(sd)
    mine = mine.to("cuda:0").train(); ref.train()
    pre, post, _ = synth.synth_batch(batch, size, seed=dseed)
    labels = synth.synth_scd_labels(batch, size, seed=dseed)
    om.scd_loss(*ref.update_scd(pre, post), labels).backward()
    o_d = mine.update_scd(pre.to("cuda:0"), post.to("cuda:0"))
    scd_loss(CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity(), o_d, labels.to("cuda:0"))[0].backward()
    torch.cuda.synchronize()
    pref = dict(ref.named_parameters())
    errs = sorted(rel(p.grad, pref[n].grad) for n, p in hot_path_named_params(mine))
    print(f"wseed {wseed} dseed {dseed}: worst {errs[-1]:.2e} median {errs[len(errs)//2]:.2e} over1e-4 {sum(e >= 1e-4 for e in errs)}", flush=True)
