import faulthandler, sys, os, torch
faulthandler.enable()
sys.path.insert(0, os.getcwd())
from change3d_amd import synthetic as synth  # deterministic synthetic weights / batches (neutral module)
from change3d_amd.synthetic import make_args
from change3d_amd.model.trainer import Trainer
from change3d_amd.model.utils import BCEDiceLoss, FusedAdam, ParamArena, hot_path_named_params
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
dev = torch.device("cuda:0")
margs = make_args(size=64); margs.act_dtype = torch.bfloat16
net = Trainer(margs); net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25)); net = net.to(dev).train()
arena = ParamArena(hot_path_named_params(net), dev)
opt = FusedAdam(arena, lr=2e-4, capturable=True)
pre, post, tgt = (t.to(dev) for t in synth.synth_batch(4, 64, seed=0))
def fwd_bwd():
    opt.zero_grad()
    prob = net.update_bcd(pre, post)
    loss = BCEDiceLoss(prob, tgt)
    loss.backward()
    return loss.detach()
opt.prepare_step()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        l = fwd_bwd(); opt.launch()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warm ok", float(l), flush=True)
g = torch.cuda.CUDAGraph()
print("capturing", flush=True)
with torch.cuda.graph(g, stream=side):
    lg = fwd_bwd()
    opt.launch()
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    opt.prepare_step(); g.replay(); torch.cuda.synchronize(); print("replay", i, float(lg), flush=True)
