"""MI355X-native mirror of the reference's `scripts/train_SCD.py` training loop (reference
scripts/train_SCD.py:181-276 `train`, :279-380 `trainValidate`).

Same loop order and loss composition
    pre_label, post_label *= label_change                                         (:216-217)
    pre_mask, post_mask, change_mask = model.update_scd(pre, post)                (:223)
    loss = 0.5 * (CE(pre_mask, pre_label) + CE(post_mask, post_label))            (:226, ignore_index=0)
         + BCEDiceLoss(change_mask, label_change)                                 (:227)
         + ChangeSimilarity(pre_mask[:, 1:], post_mask[:, 1:], label_change)      (:228)
same Adam hyper-parameters (:323-329) and the per-iteration semantic accuracy of `model/utils.py:313-319`
(`accuracy(pred * change, label)`), computed on the device.  As in `train_BCD.py` of this package the
file datasets / cv2 augmentation are out of scope: `--dataset SYNTH-SECOND` draws SECOND-shaped synthetic
pairs (K=3 perception frames, 7 classes).  Under torch.distributed.run it trains data-parallel.
"""
import os
import sys
import time
from argparse import ArgumentParser

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from change3d_amd.hostopt import freeze_gc  # noqa: E402
from change3d_amd.model.trainer import Trainer  # noqa: E402
from change3d_amd.model.utils import (AverageMeter, BCEDiceLoss, ChangeSimilarity, CrossEntropyLoss2d, FusedAdam,  # noqa: E402
                                      SCDHistogram, adjust_learning_rate)
from change3d_amd.parallel import broadcast_module_state, host_barrier, setup_data_parallel  # noqa: E402


class SyntheticSCDLoader:
    """Stand-in for the reference loader (scripts/train_SCD.py:36-96): yields (img[B,6,H,W] float32 with the
    (u8/255-0.5)/0.5 normalisation, labels[B,3,H,W] int64 = pre classes, post classes, change mask)."""

    def __init__(self, n_pairs, batch_size, size, num_class, seed):
        self.n, self.bs, self.size, self.nc, self.seed = n_pairs, batch_size, size, num_class, seed
        self.nb = n_pairs // batch_size

    def __len__(self):
        return self.nb

    def __iter__(self):
        rng = np.random.default_rng(self.seed)
        S, cell = self.size, max(self.size // 8, 1)
        for _ in range(self.nb):
            u8 = rng.integers(0, 256, size=(self.bs, 6, S, S), dtype=np.uint8)
            img = (u8.astype(np.float32) / 255.0 - 0.5) / 0.5
            grid = rng.integers(0, self.nc, size=(self.bs, 2, -(-S // cell), -(-S // cell)))
            maps = np.repeat(np.repeat(grid, cell, axis=2), cell, axis=3)[:, :, :S, :S]
            change = np.zeros((self.bs, 1, S, S), dtype=np.int64)
            for k in range(self.bs):
                for _ in range(int(rng.integers(1, 4))):
                    h, w = (int(rng.integers(S // 16, S // 4)) for _ in range(2))
                    y0, x0 = int(rng.integers(0, S - h)), int(rng.integers(0, S - w))
                    change[k, 0, y0:y0 + h, x0:x0 + w] = 1
            yield torch.from_numpy(img), torch.from_numpy(np.concatenate([maps, change], axis=1).astype(np.int64))


def scd_loss(seg_loss, sim_loss, masks, labels):
    """Loss of reference scripts/train_SCD.py:216-229; returns (loss, pre_label, post_label, label_change)."""
    pre_mask, post_mask, change_mask = masks
    label_change = labels[:, 2].long()
    pre_label, post_label = labels[:, 0].long() * label_change, labels[:, 1].long() * label_change
    segm = seg_loss(pre_mask, pre_label) + seg_loss(post_mask, post_label)
    binary = BCEDiceLoss(change_mask, label_change.unsqueeze(1).float())
    sim = sim_loss(pre_mask[:, 1:], post_mask[:, 1:], label_change.unsqueeze(1))
    return segm * 0.5 + binary + sim, pre_label, post_label, label_change


def train(args, loader, model, optimizer, sync, epoch, max_batches, cur_iter=0):
    model.train()
    seg_loss, sim_loss = CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity()
    losses, accs, lr = [], [], args.lr
    for it, (imgs, labels) in enumerate(loader):
        if it + cur_iter == 3:   # once per run, after the first iterations have built everything long-lived (hostopt.py)
            freeze_gc()
        pre, post, labels = imgs[:, 0:3].cuda().float(), imgs[:, 3:6].cuda().float(), labels.cuda()
        start = time.time()
        lr = adjust_learning_rate(args, optimizer, epoch, it + cur_iter, max_batches)
        masks = model.update_scd(pre, post)
        loss, pre_label, post_label, _ = scd_loss(seg_loss, sim_loss, masks, labels)
        optimizer.zero_grad()
        loss.backward()
        sync.finish()
        optimizer.step()
        with torch.no_grad():  # reference :241-257, on the device
            chg = (masks[2].detach() > 0.5).squeeze(1).long()
            pa, pb = masks[0].detach().argmax(1) * chg, masks[1].detach().argmax(1) * chg
            accs.append(0.5 * ((pa == pre_label).float().mean() + (pb == post_label).float().mean()))
        losses.append(loss.detach())
        if (it + 1) % 5 == 0 and args.rank == 0:
            print(f"[epoch {epoch}] [iter {it + 1}/{len(loader)}] [lr {lr:.6f}] [loss {loss.item():.4f}] "
                  f"[acc {accs[-1].item():.4f}] [{time.time() - start:.3f}s/it]")
    return float(torch.stack(losses).mean()), float(torch.stack(accs).mean()), lr


@torch.no_grad()
def val(args, val_loader, model, seg_loss=None, sim_loss=None):
    """reference scripts/train_SCD.py:104-178: eval-mode forward, the same loss composition, per-sample semantic
    accuracy (`model/utils.py:313-319` on `argmax * change` vs `label * change`, averaged over the two dates) and
    Fscd / mIoU / SeK from the joint histogram of every (prediction, label) pair (`SCDD_eval_all`, :345-378).  The
    reference moves all masks to the host every iteration; here the histogram and the accuracies are accumulated on the
    device (`c3d_hist2d`) and read back once.  Returns (Fscd, IoU_mean, Sek, acc_meter, val_loss) like the reference."""
    model.eval()
    seg_loss = seg_loss or CrossEntropyLoss2d(ignore_index=0)
    sim_loss = sim_loss or ChangeSimilarity()
    dev = next(model.parameters()).device
    hist = SCDHistogram(args.num_class, dev)
    losses, accs = [], []
    start = time.time()
    for imgs, labels in val_loader:
        pre, post, labels = imgs[:, 0:3].to(dev).float(), imgs[:, 3:6].to(dev).float(), labels.to(dev)
        masks = model.update_scd(pre, post)
        loss, pre_label, post_label, _ = scd_loss(seg_loss, sim_loss, masks, labels)
        losses.append(loss)
        chg = (masks[2] > 0.5).squeeze(1).long()
        pa, pb = masks[0].argmax(1) * chg, masks[1].argmax(1) * chg
        # accuracy(pred, label): every pixel is valid (label >= 0); per sample, mean of the two dates (:158-163)
        accs.append(0.5 * ((pa == pre_label).float().flatten(1).mean(1) + (pb == post_label).float().flatten(1).mean(1)))
        hist.update(pa, pre_label)
        hist.update(pb, post_label)
    val_loss, acc_meter = AverageMeter(), AverageMeter()
    for l in torch.stack(losses).cpu().numpy():
        val_loss.update(l)
    for a in torch.cat(accs).cpu().tolist():
        acc_meter.update(a)
    Fscd, IoU_mean, Sek = hist.scores()
    if getattr(args, "rank", 0) == 0:
        print(f"{time.time() - start:.1f}s Val loss: {val_loss.average():.2f} Fscd: {Fscd * 100:.2f} IoU: {IoU_mean * 100:.2f} "
              f"Sek: {Sek * 100:.2f} Accuracy: {acc_meter.average() * 100:.2f}")
    return Fscd, IoU_mean, Sek, acc_meter, val_loss


def trainValidate(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(args.gpu_id)))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.manual_seed(seed=16)
    torch.cuda.manual_seed(seed=16)
    args.act_dtype = torch.bfloat16 if args.act_dtype == "bf16" else torch.float32
    model = Trainer(args).cuda()
    broadcast_module_state(model)
    loader = SyntheticSCDLoader(args.synthetic_pairs, args.batch_size, args.in_height, args.num_class, seed=10 + args.rank)
    max_batches = len(loader)
    args.max_epochs = int(np.ceil(args.max_steps / max_batches))
    arena, sync = setup_data_parallel(model, torch.device("cuda", local))
    optimizer = FusedAdam(arena, args.lr, (0.9, 0.99), eps=1e-08, weight_decay=1e-4)
    cur_iter = 0
    for epoch in range(args.max_epochs):
        loss_tr, acc_tr, lr = train(args, loader, model, optimizer, sync, epoch, max_batches, cur_iter)
        cur_iter += len(loader)
        if args.rank == 0:
            print(f"Epoch {epoch}: train loss {loss_tr:.4f}  acc {acc_tr:.4f}  lr {lr:.6f}")
            if args.val_pairs > 0:   # rank 0 validates on its own BatchNorm statistics (the ones a checkpoint would hold)
                vl = SyntheticSCDLoader(args.val_pairs, min(args.batch_size, args.val_pairs), args.in_height, args.num_class, seed=5)
                val(args, vl, model)
        host_barrier()   # the other ranks wait (on the host) for rank 0's validation pass before the next exchange / teardown
    if world > 1:
        dist.destroy_process_group()


def build_parser():
    p = ArgumentParser()
    p.add_argument("--dataset", default="SYNTH-SECOND")
    p.add_argument("--in_height", type=int, default=256)
    p.add_argument("--in_width", type=int, default=256)
    p.add_argument("--num_perception_frame", type=int, default=3)
    p.add_argument("--num_class", type=int, default=7)
    p.add_argument("--max_steps", type=int, default=80000)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--lr", type=float, default=2e-4)
    p.add_argument("--lr_mode", default="poly")
    p.add_argument("--step_loss", type=int, default=100)
    p.add_argument("--pretrained", default="./pretrained/X3D_L.pyth")
    p.add_argument("--gpu_id", default=0, type=int)
    p.add_argument("--synthetic_pairs", type=int, default=256)
    p.add_argument("--val_pairs", type=int, default=0, help="validate on this many synthetic pairs after every epoch")
    p.add_argument("--act_dtype", choices=["f32", "bf16"], default="bf16")
    return p


if __name__ == "__main__":
    trainValidate(build_parser().parse_args())
