"""smoke(): one tiny BCD train step (forward, BCE+Dice, backward, fused Adam) on cuda:0 through
the HIP kernels, checked against the CPU oracle (oracle/ is test infrastructure: used here only
as the checker)."""
import torch


def smoke(size=64, batch=2, verbose=True):
    from oracle import model as om, synth
    from .model.trainer import Trainer
    from .model.utils import BCEDiceLoss, FusedAdam, ParamArena, hot_path_named_params

    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    dev = torch.device("cuda:0")
    args = om.make_args(size=size)
    ref = om.Trainer(args)
    sd = synth.synth_state_dict(ref, seed=16, mask_margin=0.25)
    ref.load_state_dict(sd)
    ref.train()
    pre, post, tgt = synth.synth_batch(batch, size, seed=0)
    p_ref = ref.update_bcd(pre, post)
    l_ref = om.bce_dice_loss(p_ref, tgt)
    l_ref.backward()

    net = Trainer(args)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    arena = ParamArena(hot_path_named_params(net), dev)
    opt = FusedAdam(arena, lr=2e-4)
    opt.zero_grad()
    prob = net.update_bcd(pre.to(dev), post.to(dev))
    loss = BCEDiceLoss(prob, tgt.to(dev))
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    dp = (prob.detach().cpu() - p_ref.detach()).abs().max().item()
    dl = abs(loss.item() - l_ref.item())
    gref = dict(ref.named_parameters())
    worst = 0.0
    for n, p in hot_path_named_params(net):
        g = gref[n].grad
        rel = (p.grad.cpu() - g).norm().item() / (g.norm().item() + 1e-12)
        worst = max(worst, rel)
    if verbose:
        print(f"[smoke] max|dprob|={dp:.3e} |dloss|={dl:.3e} worst rel grad err={worst:.3e}")
    assert dp < 1e-4 and dl < 1e-4 and worst < 2e-3, (dp, dl, worst)
    return dp, dl, worst
