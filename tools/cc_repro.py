"""Diagnostic: run-to-run reproducibility of the CC encoder backward, stage by stage (gradient of every stage output)."""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_bf16_fullsize_scd_cc_gpu as T
from change3d_amd import synthetic as synth
from change3d_amd.model.caption_decoder import packed_cross_entropy
from change3d_amd.model.utils import ParamArena, cc_named_params
B, S = int(os.environ.get("B", 4)), int(os.environ.get("S", 128))
T.B, T.S = B, S
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
pre, post, _ = (t.to("cuda:0") for t in synth.synth_batch(B, S, seed=0))
caps, caplens = (t.to("cuda:0") for t in synth.synth_captions(B, seed=0, vocab_size=501))
net = T._build_cc(dt)
enc_named, dec_named = cc_named_params(net)
arenas = (ParamArena(enc_named, torch.device("cuda:0")), ParamArena(dec_named, torch.device("cuda:0")))
st0 = {k: v.clone() for k, v in net.state_dict().items()}
runs = []
for r in range(2):
    net.load_state_dict(st0)
    grads = {}
    hooks = []
    for i in range(5):
        def mk(i):
            def fwd_hook(m, inp, out):
                out.register_hook(lambda g: grads.__setitem__(f"d_out_blocks{i}", g.detach().float().clone()))
            return fwd_hook
        hooks.append(net.encoder.x3d.blocks[i].register_forward_hook(mk(i)))
    for a in arenas:
        a.zero_grad()
    feat = net.update_cc(pre, post)
    feat.register_hook(lambda g: grads.__setitem__("d_feat", g.detach().float().clone()))
    Bc, Cc, Hc, Wc = feat.shape
    logits = net.decoder.logits_seq_first(feat.permute(2, 3, 0, 1).reshape(Hc * Wc, Bc, Cc), caps)
    packed_cross_entropy(logits, caps, caplens, 501, ignore_index=0).backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    runs.append(grads)
for k in ["d_feat"] + [f"d_out_blocks{i}" for i in (4, 3, 2, 1, 0)]:
    a, b = runs[0][k].double(), runs[1][k].double()
    print(f"{k:16s} |g| {a.norm().item():.4e}  run-to-run rel {((a - b).norm() / (a.norm() + 1e-30)).item():.3e}  max|d| {(a - b).abs().max().item():.3e}")
