"""MI355X-native mirror of the hot-path pieces of the reference's `model/utils.py`:
`weight_init` (reference model/utils.py:20-82), `adjust_learning_rate` (:84-152),
`BCEDiceLoss` (:154-169) — the loss runs as HIP reduction kernels — plus the pieces the
data-parallel train step needs that the reference gets from torch.optim: a flat parameter /
gradient arena and a fused Adam (`scripts/train_BCD.py:284-290` hyper-parameters).
"""
import math
import os

import numpy as np

import torch
import torch.nn as nn

from .. import ops


def weight_init(module):
    """Conv2d (direct child or inside a Sequential) -> kaiming-normal fan_in/relu, norm layers
    -> 1/0, Linear -> kaiming-normal; ConvTranspose2d is not matched and keeps torch's default
    init (reference quirk)."""
    for _, child in module.named_children():
        if isinstance(child, (nn.AdaptiveAvgPool2d, nn.AdaptiveMaxPool2d, nn.ModuleList, nn.BCELoss)):
            continue
        if isinstance(child, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(child.weight, mode="fan_in", nonlinearity="relu")
            if child.bias is not None:
                nn.init.zeros_(child.bias)
        elif isinstance(child, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.ones_(child.weight)
            if child.bias is not None:
                nn.init.zeros_(child.bias)
        elif isinstance(child, nn.Sequential):
            for _, sub in child.named_children():
                if isinstance(sub, (nn.Conv2d, nn.Linear)):
                    nn.init.kaiming_normal_(sub.weight, mode="fan_in", nonlinearity="relu")
                    if sub.bias is not None:
                        nn.init.zeros_(sub.bias)
                elif isinstance(sub, (nn.BatchNorm2d, nn.GroupNorm)):
                    nn.init.ones_(sub.weight)
                    if sub.bias is not None:
                        nn.init.zeros_(sub.bias)
                else:
                    weight_init(sub)
        elif len(list(child.children())) > 0:
            weight_init(child)


def adjust_learning_rate(args, optimizer, epoch=None, iter=None, max_batches=None, lr_factor=1.0,
                         shrink_factor=None, verbose=True):
    """Same schedule and argument meaning as reference model/utils.py:84-152."""
    if shrink_factor is not None:
        if not 0 < shrink_factor < 1:
            raise ValueError(f"Shrink factor must be between 0 and 1, got {shrink_factor}")
        if verbose:
            print("\nDECAYING learning rate.")
        for group in optimizer.param_groups:
            group["lr"] = group["lr"] * shrink_factor
        if verbose:
            print(f"The new learning rate is {optimizer.param_groups[0]['lr']:.6f}\n")
        return optimizer.param_groups[0]["lr"]
    if args.lr_mode == "step":
        if epoch is None:
            raise ValueError("Epoch must be provided for step lr_mode")
        lr = args.lr * (0.1 ** (epoch // args.step_loss))
    elif args.lr_mode == "poly":
        if any(p is None for p in [epoch, iter, max_batches]):
            raise ValueError("Epoch, iter, and max_batches must be provided for poly lr_mode")
        max_iter = max_batches * args.max_epochs
        lr = args.lr * (1 - iter * 1.0 / max_iter) ** 0.9
    else:
        raise ValueError(f"Unknown lr mode {args.lr_mode}")
    if epoch == 0 and iter is not None and iter < 200:
        lr = args.lr * 0.9 * (iter + 1) / 200 + 0.1 * args.lr
    lr *= lr_factor
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def load_checkpoint(args, model, save_path, max_batches):
    """Resume contract of reference model/utils.py:205-232: `<save_path>/checkpoint.pth.tar` holds
    {'epoch', 'state_dict', ...}; only the weights and the epoch are restored (never the optimizer state).
    Returns (start_epoch, cur_iter).  `map_location='cpu'`: `load_state_dict` copies into the parameters'
    own (arena) storage, so views held by ParamArena / kernels stay valid."""
    start_epoch, cur_iter = 0, 0
    if args.resume is not None:
        checkpoint_path = os.path.join(save_path, "checkpoint.pth.tar")
        if os.path.isfile(checkpoint_path):
            print(f"=> loading checkpoint '{checkpoint_path}'")
            checkpoint = torch.load(checkpoint_path, map_location="cpu")
            start_epoch = checkpoint["epoch"]
            cur_iter = start_epoch * max_batches
            model.load_state_dict(checkpoint["state_dict"])
            print(f"=> loaded checkpoint '{checkpoint_path}' (epoch {checkpoint['epoch']})")
        else:
            print(f"=> no checkpoint found at '{checkpoint_path}'")
    return start_epoch, cur_iter


_LOG_HEADERS = {
    ("LEVIR-CD", "WHU-CD", "CLCD"): ("Epoch", "Kappa (val)", "IoU (val)", "F1 (val)", "R (val)", "P (val)"),
    ("HRSCD", "SECOND"): ("epoch", "train_loss", "train_acc", "val_Fscd", "val_IoU_mean", "val_Sek", "val_loss", "val_acc"),
    ("xBD",): ("epoch", "loss_val", "loc_f1_score", "harmonic_mean_f1", "oa_f1", "damage_f1_scores"),
    ("LEVIR-CC", "DUBAI-CC"): ("epoch", "loss_val", "loc_f1_score", "harmonic_mean_f1", "oa_f1", "damage_f1_scores"),
}


def setup_logger(args, save_path):
    """Log-file contract of reference model/utils.py:235-276: append the argument dump and the per-dataset
    column header to `<save_path>/<args.log_file>`, return the open handle."""
    logger = open(os.path.join(save_path, args.log_file), "a+")
    logger.write("Model Configurations:\n")
    for arg, value in vars(args).items():
        logger.write(f"{arg}: {value}\n")
        print(f"{arg}: {value}")
    logger.write("\n" + "-" * 60)
    for names, cols in _LOG_HEADERS.items():
        if args.dataset in names:
            logger.write("\n" + "\t".join(cols))
            break
    else:
        assert False, r"setup_logger error, please check the input dataset!"
    logger.flush()
    return logger


class _BCEDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets):
        ops.require_gpu(inputs, "loss input")
        p = inputs.detach().contiguous().float()
        t = targets.detach().contiguous().float()
        sums = torch.empty(4, dtype=torch.float64, device=p.device)
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        ops.bce_dice_fwd(p, t, sums, loss)
        ctx.saved = (p, t, sums)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        p, t, sums = ctx.saved
        dp = torch.empty_like(p)
        dl = dloss.detach().reshape(1).contiguous().float()
        ops.bce_dice_bwd(p, t, sums, dl, dp)
        return dp, None


def BCEDiceLoss(inputs, targets):
    """bce(mean) + 1 - (2*sum(p*t)+1e-5)/(sum(p)+sum(t)+1e-5), sums over the whole batch."""
    return _BCEDiceFn.apply(inputs, targets)


def _pixel_contiguous(x):
    """f32 NCHW view with contiguous pixels (a channel slice such as `mask[:, 1:]` qualifies)."""
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    B, C, H, W = x.shape
    if (W > 1 and x.stride(3) != 1) or (H > 1 and x.stride(2) != W):
        x = x.contiguous()
    return x


class _CE2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, ignore_index):
        ops.require_gpu(inputs, "loss input")
        x = _pixel_contiguous(inputs)
        t = targets.detach().contiguous().long()
        sums = torch.empty(2, dtype=torch.float64, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        ops.ce2d_fwd(x, t, ignore_index, sums, loss)
        ctx.saved, ctx.ignore_index = (x, t, sums), ignore_index
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        x, t, sums = ctx.saved
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        ops.ce2d_bwd(x, t, ctx.ignore_index, sums, dloss.detach().reshape(1).contiguous().float(), dx)
        return dx, None, None


class CrossEntropyLoss2d(nn.Module):
    """reference model/utils.py:171-178: `nll_loss(log_softmax(inputs, 1), targets, ignore_index, 'mean')` as
    one fused HIP pass (scripts/train_SCD.py:226 uses ignore_index=0)."""

    def __init__(self, weight=None, ignore_index=-1):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("class weights are not used by any Change3D path")
        self.ignore_index = ignore_index

    def forward(self, inputs, targets):
        return _CE2dFn.apply(inputs, targets, self.ignore_index)


class _ChangeSimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, label_change):
        ops.require_gpu(x1, "loss input")
        a, b = _pixel_contiguous(x1), _pixel_contiguous(x2)
        lc = label_change.detach().reshape(a.shape[0], -1).contiguous().long()
        sums = torch.empty(1, dtype=torch.float64, device=a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        ops.cossim_fwd(a, b, lc, sums, loss)
        ctx.saved = (a, b, lc)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        a, b, lc = ctx.saved
        da = torch.empty(a.shape, dtype=torch.float32, device=a.device)
        db = torch.empty(b.shape, dtype=torch.float32, device=a.device)
        ops.cossim_bwd(a, b, lc, dloss.detach().reshape(1).contiguous().float(), da, db)
        return da, db, None


class ChangeSimilarity(nn.Module):
    """reference model/utils.py:180-203: cosine-embedding loss between the per-pixel class distributions of
    the two semantic heads (+1 where unchanged, -1 where changed), softmax fused in."""

    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("the reference only uses reduction='mean'")

    def forward(self, x1, x2, label_change):
        return _ChangeSimFn.apply(x1, x2, label_change)


# ----------------------------------------------------------------------------- arenas + Adam
class ParamArena:
    """One flat f32 buffer for parameters and one for their gradients.

    Parameters keep their identity (names, shapes, state-dict behaviour) but their storage
    becomes a view into `flat_param`, `p.grad` a view into `flat_grad`: kernels accumulate
    gradients in place, Adam is one launch, and data-parallel training all-reduces ONE
    contiguous buffer.  Only parameters that take part in the hot path are
    placed here (the never-executed res5/head of BCD are left alone, their grad stays None,
    exactly like the reference where Adam skips them)."""

    def __init__(self, named_params, device):
        self.names, self.params, self.offsets = [], [], []
        off = 0
        for name, p in named_params:
            self.names.append(name)
            self.params.append(p)
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def attach_grads(self):
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def check_grads_attached(self):
        """Raise if a parameter's `.grad` is no longer the arena view.  `model.zero_grad()` / `p.grad = None`
        (torch's set_to_none default) detach it: the kernels would then accumulate into a fresh tensor while
        Adam and the all-reduce read the stale flat buffer, and training would silently do nothing.  Clear
        gradients with `optimizer.zero_grad()` (FusedAdam) or `arena.zero_grad()`."""
        base = self.flat_grad.data_ptr()
        for n, p, o in zip(self.names, self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                raise RuntimeError(
                    f"gradient of '{n}' is not a view of the flat gradient arena (was model.zero_grad() or "
                    f"p.grad = None used?): clear gradients with FusedAdam.zero_grad() / ParamArena.zero_grad()")

    def zero_grad(self):
        self.flat_grad.zero_()
        self.attach_grads()


class FusedAdam:
    """torch.optim.Adam semantics (coupled L2 weight decay) over a ParamArena, one HIP kernel
    per step.  `param_groups` exists so `adjust_learning_rate` works unchanged."""

    def __init__(self, arena, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4, capturable=False):
        self.arena = arena
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, params=arena.params)]
        dev = arena.flat_param.device
        self.exp_avg = torch.zeros_like(arena.flat_param)
        self.exp_avg_sq = torch.zeros_like(arena.flat_param)
        self.step_count = 0
        self.capturable = capturable
        self.hp_dev = torch.zeros(3, dtype=torch.float32, device=dev) if capturable else None

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def hparams(self, step):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        return float(g["lr"]), float(bc1), float(math.sqrt(bc2))

    def prepare_step(self):
        """Host-side part of a step (call OUTSIDE a captured graph, before replay)."""
        self.arena.check_grads_attached()
        self.step_count += 1
        lr, bc1, bc2s = self.hparams(self.step_count)
        if self.capturable:
            self.hp_dev.copy_(torch.tensor([lr, bc1, bc2s], dtype=torch.float32), non_blocking=True)
        return lr, bc1, bc2s

    def launch(self, lr=0.0, bc1=1.0, bc2s=1.0):
        ops.bump_weights_version()   # in-place parameter update: folded-BatchNorm copies made earlier are stale
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        a = self.arena
        ops.adam_step(a.flat_param, a.flat_grad, self.exp_avg, self.exp_avg_sq, a.numel, self.hp_dev, lr, bc1, bc2s,
                      b1, b2, g["eps"], g["weight_decay"])

    def step(self):
        lr, bc1, bc2s = self.prepare_step()
        self.launch(lr, bc1, bc2s)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}


def hot_path_named_params(trainer):
    """Parameters that receive gradients on the BCD/SCD/BDA path: everything except the
    never-executed `encoder.x3d.blocks.4` (res5) and `.blocks.5` (head)."""
    skip = ("encoder.x3d.blocks.4.", "encoder.x3d.blocks.5.")
    return [(n, p) for n, p in trainer.named_parameters() if not n.startswith(skip)]


def cc_named_params(trainer):
    """Parameters that receive gradients on the change-captioning path (reference scripts/train_CC.py:436-458 builds one
    Adam over `trainer.encoder.parameters()` and one over `trainer.decoder.parameters()`; parameters whose `.grad` stays
    None -- the X3D head, `encoder.fc.*`, and the decoder modules the layer never runs -- are skipped by Adam there and
    are simply not placed in the arenas here).  Returns (encoder_named, decoder_named)."""
    enc = [(n, p) for n, p in trainer.named_parameters()
           if n.startswith("encoder.") and not n.startswith(("encoder.x3d.blocks.5.", "encoder.fc."))]
    used = {id(p) for p in trainer.decoder.used_parameters()}
    dec = [(n, p) for n, p in trainer.named_parameters() if n.startswith("decoder.") and id(p) in used]
    return enc, dec


def clip_gradient(optimizer, grad_clip):
    """reference model/utils.py:481-491: clamp every gradient to [-grad_clip, grad_clip]; for a FusedAdam over a
    ParamArena that is ONE HIP pass over the flat gradient buffer."""
    arena = getattr(optimizer, "arena", None)
    if arena is not None and arena.flat_grad.is_cuda:
        ops.clamp_(arena.flat_grad, grad_clip)
        return
    for group in optimizer.param_groups:
        for param in group["params"]:
            if param.grad is not None:
                param.grad.data.clamp_(-grad_clip, grad_clip)


# ------------------------------------------------------------------------------ validation metrics (SCD / CC scripts)
class AverageMeter:
    """reference model/utils.py:278-310 (imported by scripts/train_SCD.py:22-32 and scripts/train_CC.py:22-23):
    `average()` = sum(val * weight) / sum(count)."""

    def __init__(self):
        self.initialized, self.val, self.avg, self.sum, self.count = False, None, None, None, None

    def initialize(self, val, count, weight):
        self.val, self.avg, self.count, self.sum, self.initialized = val, val, count, val * weight, True

    def update(self, val, count=1, weight=1):
        if not self.initialized:
            self.initialize(val, count, weight)
        else:
            self.add(val, count, weight)

    def add(self, val, count, weight):
        self.val = val
        self.count += count
        self.sum += val * weight
        self.avg = self.sum / self.count

    def value(self):
        return self.val

    def average(self):
        return self.avg


def accuracy(pred, label, ignore_zero=False):
    """reference model/utils.py:313-319: fraction of the valid pixels (label >= 0, or > 0) where pred == label, and the
    number of valid pixels.  numpy arrays or torch tensors (device tensors stay on the device: one read-back of two
    scalars)."""
    valid = (label > 0) if ignore_zero else (label >= 0)
    acc_sum = (valid * (pred == label)).sum()
    valid_sum = valid.sum()
    if isinstance(valid_sum, torch.Tensor):
        acc_sum, valid_sum = (int(v) for v in torch.stack([acc_sum, valid_sum]).tolist())
    return float(acc_sum) / (valid_sum + 1e-10), valid_sum


class SCDHistogram:
    """Joint (prediction x label) histogram of the SCD validation loop (reference model/utils.py:321-355: fast_hist /
    get_hist summed over every (pred, label) pair of the epoch), accumulated ON THE DEVICE by `c3d_hist2d`: the reference
    moves every mask to the host each iteration; here one n*n read-back happens in `scores()`."""

    def __init__(self, num_class, device):
        self.n = int(num_class)
        self.hist = torch.zeros(self.n * self.n + 1, dtype=torch.int64, device=device)

    def update(self, pred, label):
        pred, label = pred.reshape(-1).to(torch.int64).contiguous(), label.reshape(-1).to(torch.int64).contiguous()
        if pred.numel() != label.numel():
            raise AssertionError("The size of prediction and target must be the same")
        ops.hist2d(pred, label, self.n, self.hist)

    def matrix(self):
        h = self.hist.cpu().numpy()
        if h[-1] != 0:
            raise ValueError(f"{int(h[-1])} label values outside [0, {self.n}) in the SCD validation histogram")
        return h[:-1].reshape(self.n, self.n).astype(np.float64)

    def scores(self):
        return scd_scores_from_hist(self.matrix())


def _cal_kappa(hist):
    """reference model/utils.py:330-342."""
    if hist.sum() == 0:
        return 0
    po = np.diag(hist).sum() / hist.sum()
    pe = np.matmul(hist.sum(1), hist.sum(0).T) / hist.sum() ** 2
    return 0 if pe == 1 else (po - pe) / (1 - pe)


def scd_scores_from_hist(hist):
    """(Fscd, mIoU, SeK) from the num_class x num_class histogram: reference model/utils.py:356-378 (float64 host
    arithmetic, same order of operations)."""
    hist = np.asarray(hist, dtype=np.float64)
    c2 = np.zeros((2, 2))
    c2[0][0] = hist[0][0]
    c2[0][1] = hist.sum(1)[0] - hist[0][0]
    c2[1][0] = hist.sum(0)[0] - hist[0][0]
    c2[1][1] = hist[1:, 1:].sum()
    hist_n0 = hist.copy()
    hist_n0[0][0] = 0
    kappa_n0 = _cal_kappa(hist_n0)
    iu = np.diag(c2) / (c2.sum(1) + c2.sum(0) - np.diag(c2))
    sek = (kappa_n0 * math.exp(iu[1])) / math.e
    pixel_sum = hist.sum()
    change_pred_sum = pixel_sum - hist.sum(1)[0].sum()
    change_label_sum = pixel_sum - hist.sum(0)[0].sum()
    sc_tp = np.diag(hist[1:, 1:]).sum()
    precision, recall = sc_tp / change_pred_sum, sc_tp / change_label_sum
    fscd = 2.0 / (1.0 / precision + 1.0 / recall) if precision > 0 and recall > 0 else 0.0   # scipy.stats.hmean of the two
    return fscd, (iu[0] + iu[1]) / 2, sek


def SCDD_eval_all(preds, labels, num_class):
    """reference model/utils.py:345-378: lists of per-image prediction / label maps -> (Fscd, mIoU, SeK).  Device tensors
    are histogrammed by `c3d_hist2d`; numpy arrays (the reference's calling convention) by np.bincount."""
    if len(preds) and isinstance(preds[0], torch.Tensor) and preds[0].is_cuda:
        acc = SCDHistogram(num_class, preds[0].device)
        for p, l in zip(preds, labels):
            if tuple(p.shape) != tuple(l.shape):
                raise AssertionError("The size of prediction and target must be the same")
            acc.update(p, l)
        return acc.scores()
    hist = np.zeros((num_class, num_class))
    for p, l in zip(preds, labels):
        p, l = np.asarray(p), np.asarray(l)
        assert set(np.unique(p)).issubset({0, 1, 2, 3, 4, 5, 6}), "unrecognized label number"
        assert p.shape == l.shape, "The size of prediction and target must be the same"
        a, b = p.flatten(), l.flatten()
        k = (a >= 0) & (a < num_class)
        hist += np.bincount(num_class * a[k].astype(int) + b[k], minlength=num_class ** 2).reshape(num_class, num_class)
    return scd_scores_from_hist(hist)


def caption_accuracy(scores, targets, k):
    """reference model/utils.py:493-507: top-k accuracy (percent) of packed (rows, vocab) scores against (rows,) targets."""
    batch_size = targets.size(0)
    _, ind = scores.topk(k, 1, True, True)
    correct = ind.eq(targets.view(-1, 1).expand_as(ind))
    return correct.view(-1).float().sum().item() * (100.0 / batch_size)


def eval_caption_score(references, hypotheses):
    """reference model/utils.py:509-536: BLEU-1..4 / METEOR / ROUGE-L / CIDEr of decoded captions through pycocoevalcap (its METEOR
    scorer drives a java process).  Host-side string scoring, outside the MI355X hot path (out of scope):
    where pycocoevalcap is installed this delegates to it with the reference's argument handling and return value (so a ported
    `scripts/train_CC.py` validation pass completes); where it is not, it raises -- no silent substitute score."""
    try:
        from pycocoevalcap.bleu.bleu import Bleu
        from pycocoevalcap.cider.cider import Cider
        from pycocoevalcap.meteor.meteor import Meteor
        from pycocoevalcap.rouge.rouge import Rouge
    except ImportError as e:
        raise NotImplementedError("caption text metrics need pycocoevalcap (BLEU / METEOR / ROUGE-L / CIDEr; host-side string scoring, "
                                  "not part of the MI355X hot path) and it is not importable here: " + str(e)) from e
    scorers = [(Bleu(4), ["Bleu_1", "Bleu_2", "Bleu_3", "Bleu_4"]), (Meteor(), "METEOR"), (Rouge(), "ROUGE_L"), (Cider(), "CIDEr")]
    hypo = [[" ".join(str(t) for t in h)] for h in hypotheses]
    ref = [[" ".join(str(t) for t in r) for r in rs] for rs in references]
    out = {}
    for scorer, name in scorers:
        value, _ = scorer.compute_score(ref, hypo)
        print("{} {}".format(name, value))
        out.update(zip(name, value) if isinstance(name, list) else [(name, value)])
    return out
