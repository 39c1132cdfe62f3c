import contextlib, io, sys, torch
sys.path.insert(0, '.')
from change3d_amd import ops, synthetic as synth
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import BCEDiceLoss
from change3d_amd.model.x3d import X3DResStage
DEV='cuda:0'
def run(flags, opt=3):
    ops.set_option(ops.OPT_FUSE_WGRAD, opt)
    args = synth.make_args(size=64, act_dtype=torch.bfloat16)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=5, mask_margin=0.25, branch_gain=0.1))
    net = net.to(DEV).train()
    for m in net.modules():
        if isinstance(m, X3DResStage): m.driver_flags = flags
    pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(3, 64, seed=2))
    loss = BCEDiceLoss(net.update_bcd(pre, post), tgt); loss.backward(); torch.cuda.synchronize()
    return {n: p.grad.double().cpu() for n, p in net.named_parameters() if p.grad is not None}
def cmp(a, b, tag):
    r = sorted(((((a[n]-b[n]).norm()/a[n].norm().clamp_min(1e-30)).item(), n) for n in a), reverse=True)
    print(tag, 'top:', [(f'{e:.1e}', n[-40:]) for e, n in r[:6]], 'median', f'{r[len(r)//2][0]:.1e}')
base = run(0); again = run(0); sep = run(ops.STAGE_SEPARATE_WGRAD); a_only = run(0, 1); c_only = run(0, 2)
cmp(base, again, 'fused vs fused again  ')
cmp(base, sep, 'fused vs separate     ')
cmp(base, a_only, 'fused vs conv_a-only  ')
cmp(base, c_only, 'fused vs conv_c-only  ')
cmp(sep, a_only, 'separate vs conv_a-only')
