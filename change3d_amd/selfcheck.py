"""smoke(): one tiny BCD train step (forward, BCE+Dice, backward, fused Adam) on cuda:0 through
the HIP kernels, checked against the CPU oracle (oracle/ is test infrastructure: used here only
as the checker).

Tolerances follow tests/test_model_gpu.py: this 55-block train-mode-BN network with synthetic weights
is ill-conditioned -- the fp32 CPU oracle itself is 1e-4 (probabilities) and up to a few percent
(some gradients) away from its own fp64 evaluation -- so the HIP result is required to sit inside
K_NOISE x the fp32 oracle's distance from fp64, not within a fixed epsilon of the fp32 oracle."""
import torch

K_NOISE = 4.0


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
    ref64 = om.Trainer(args)
    ref64.load_state_dict(sd)
    ref64 = ref64.double().train()
    p64 = ref64.update_bcd(pre.double(), post.double())
    l64 = om.bce_dice_loss(p64, tgt.double())
    l64.backward()

    net = Trainer(args)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    arena = ParamArena(hot_path_named_params(net), dev)
    opt = FusedAdam(arena, lr=2e-4)
    opt.zero_grad()
    prob = net.update_bcd(pre.to(dev), post.to(dev))
    loss = BCEDiceLoss(prob, tgt.to(dev))
    loss.backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in hot_path_named_params(net)}
    opt.step()
    torch.cuda.synchronize()

    p64 = p64.detach()
    e_hip = (prob.detach().cpu().double() - p64).abs().max().item()
    e_ref = (p_ref.detach().double() - p64).abs().max().item()
    dl_hip, dl_ref = abs(loss.item() - l64.item()), abs(l_ref.item() - l64.item())
    g32, g64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
    rel = lambda a, b: (a - b).norm().item() / (b.norm().item() + 1e-30)  # noqa: E731
    names = list(grads)
    eh = torch.tensor([rel(grads[n], g64[n].grad) for n in names])
    er = torch.tensor([rel(g32[n].grad.double(), g64[n].grad) for n in names])
    lim = torch.maximum(torch.maximum(K_NOISE * er, K_NOISE * er.quantile(0.9)), torch.tensor(2e-3, dtype=er.dtype))
    outliers = int((eh > lim).sum())
    if verbose:
        print(f"[smoke] max|p - p_fp64|: hip {e_hip:.3e} (fp32 oracle {e_ref:.3e})  |loss - loss_fp64|: hip {dl_hip:.2e} "
              f"(oracle {dl_ref:.2e})  grad rel-L2 vs fp64: hip median {eh.median():.2e} max {eh.max():.2e} "
              f"(oracle median {er.median():.2e} max {er.max():.2e}), outliers {outliers}/{len(names)}")
    assert e_hip <= K_NOISE * e_ref + 1e-6, (e_hip, e_ref)
    assert dl_hip <= K_NOISE * dl_ref + 1e-5, (dl_hip, dl_ref)
    assert eh.median() <= 1.5 * er.median() + 1e-4 and eh.max() <= K_NOISE * er.max() + 2e-3, (eh.median(), eh.max())
    assert outliers <= 0.02 * len(names), outliers
    assert all(torch.isfinite(p).all() for _, p in hot_path_named_params(net)), "non-finite parameter after Adam"
    return e_hip, dl_hip, float(eh.max())
