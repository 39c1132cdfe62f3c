#!/usr/bin/env python
"""Is the launch stream ever waiting for the HOST inside a BCD train step?  Every kernel launch of one step (the ctypes
wrappers of change3d_amd.ops and the stage-driver calls) is preceded by a timing event; after the step, the time the GPU
reached each event is compared with the host time at which the launch was made (both relative to a synchronised start).
lead = gpu_time - host_time: large = the host is ahead (the launch sat in the queue), ~0 = the GPU waited for this launch.
usage: python tools/host_gpu_lag.py [steady|synced]   (steady: the step is enqueued behind an unfinished previous step)"""
import contextlib, io, sys, time
import torch
sys.path.insert(0, '.')
from change3d_amd import ops, synthetic as synth
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import BCEDiceLoss, FusedAdam, adjust_learning_rate
from change3d_amd.parallel import setup_data_parallel
DEV = torch.device('cuda', 0)
mode = sys.argv[1] if len(sys.argv) > 1 else 'synced'
args = synth.make_args(size=256)
args.act_dtype = torch.bfloat16
args.lr_mode, args.lr, args.max_epochs, args.step_loss = "poly", 2e-4, 1, 100
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    net = Trainer(args)
net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
net = net.to(DEV).train()
arena, sync = setup_data_parallel(net, DEV, overlap=True)
opt = FusedAdam(arena, lr=args.lr, capturable=True)
pre, post, tgt = (t.to(DEV) for t in synth.synth_batch(32, 256, seed=0))
it = [0]
def step():
    adjust_learning_rate(args, opt, 0, it[0], 80000); opt.prepare_step()
    opt.zero_grad()
    loss = BCEDiceLoss(net.update_bcd(pre, post), tgt); loss.backward(); sync.finish(); opt.launch(); it[0] += 1
for _ in range(30): step()
torch.cuda.synchronize()
log = []
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def g(*a, **k):
        ev = torch.cuda.Event(enable_timing=True); ev.record()
        log.append((label or (a[0] if name == '_launch' else name), time.perf_counter(), ev))
        return f(*a, **k)
    setattr(mod, name, g)
wrap(ops, '_launch'); wrap(ops, 'stage_fwd'); wrap(ops, 'stage_bwd')
if mode == 'steady':
    step()            # the probed step is enqueued while this one still runs
e0 = torch.cuda.Event(enable_timing=True)
log.clear()
if mode != 'steady': torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record()
step()
t_end = time.perf_counter()
torch.cuda.synchronize()
print(f"mode {mode}: host enqueue of the step {1e3*(t_end-t0):.2f} ms")
print(f"{'launch':28s} {'host ms':>9s} {'gpu ms':>9s} {'lead ms':>9s}")
for name, th, ev in log:
    tg = e0.elapsed_time(ev)
    print(f"{str(name)[:28]:28s} {1e3*(th-t0):9.3f} {tg:9.3f} {tg-1e3*(th-t0):9.3f}")
