"""MI355X-native mirror of the reference's `scripts/train_BCD.py` (reference
scripts/train_BCD.py:92-154 `val`, :157-237 `train`, :240-383 `trainValidate`, :386-485 flags).

Same flags, same loop order (LR update -> update_bcd -> BCEDiceLoss -> binarise ->
zero_grad/backward/step -> loss/metrics), same Adam hyper-parameters, checkpoint layout and
"validate on the test split, skip epoch 0" behaviour.  Differences, all deliberate:
  * file datasets / cv2 augmentation are out of scope: `--dataset SYNTH-CD`
    (default) draws LEVIR-CD-shaped synthetic pairs with the reference's normalisation;
  * the model is `change3d_amd.model.Trainer` (HIP kernels), the optimizer the fused Adam;
  * the per-iteration confusion matrix is accumulated on the GPU (4 integers come back per
    epoch instead of two label maps per iteration) and `torch.cuda.empty_cache()` is not called;
  * launched under torch.distributed.run it trains data-parallel (one rank per GPU, RCCL).
"""
import os
import sys
import time
from argparse import ArgumentParser
from os.path import join as osp

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from change3d_amd.hostopt import freeze_gc  # noqa: E402
from change3d_amd.data.transforms import DeviceBatchTransform, draw_augmentation_flags  # noqa: E402
from change3d_amd.model.trainer import Trainer  # noqa: E402
from change3d_amd.model.utils import BCEDiceLoss, FusedAdam, adjust_learning_rate  # noqa: E402
from change3d_amd.parallel import broadcast_module_state, host_barrier, setup_data_parallel  # noqa: E402
from change3d_amd.utils.metric_tool import ConfuseMatrixMeter  # noqa: E402


class SyntheticBCDLoader:
    """Stand-in for the reference DataLoader (reference scripts/train_BCD.py:31-89).  It yields what the reference's
    dataset holds BEFORE its per-sample numpy transforms -- raw uint8 pairs (B, H, W, 6), uint8 labels {0, 255}
    (B, H, W) and the random_flip / random_exchange draws -- and `DeviceBatchTransform` (change3d_amd/data/transforms.py,
    kernel c3d_bcd_preprocess) turns a batch into the normalised float tensors on the GPU in one pass
    (reference data/transforms.py:100-154 does the same per sample on the host)."""

    def __init__(self, n_pairs, batch_size, size, seed, drop_last=False, train=False):
        self.n, self.bs, self.size, self.seed, self.train = n_pairs, batch_size, size, seed, train
        self.nb = n_pairs // batch_size if drop_last else -(-n_pairs // batch_size)
        self.transform = None

    def __len__(self):
        return self.nb

    def __iter__(self):
        rng = np.random.default_rng(self.seed)
        for i in range(self.nb):
            b = min(self.bs, self.n - i * self.bs)
            u8 = rng.integers(0, 256, size=(b, self.size, self.size, 6), dtype=np.uint8)
            lab = np.zeros((b, self.size, self.size), dtype=np.uint8)
            for k in range(b):
                for _ in range(int(rng.integers(1, 4))):
                    h, w = (int(rng.integers(self.size // 16, self.size // 4)) for _ in range(2))
                    y0, x0 = int(rng.integers(0, self.size - h)), int(rng.integers(0, self.size - w))
                    lab[k, y0:y0 + h, x0:x0 + w] = 255
            flags = draw_augmentation_flags(b, rng, train=self.train)
            if self.transform is None:
                self.transform = DeviceBatchTransform(torch.device("cuda", torch.cuda.current_device()))
            pre, post, target = self.transform(u8, lab, flags)       # device tensors: (B,3,H,W) x2, (B,1,H,W)
            yield torch.cat([pre, post], dim=1), target


def create_data_loaders(args, rank=0):
    train = SyntheticBCDLoader(args.synthetic_pairs, args.batch_size, args.in_height, seed=10 + rank, drop_last=True, train=True)
    val = SyntheticBCDLoader(max(args.batch_size, args.synthetic_pairs // 8), args.batch_size, args.in_height, seed=5)
    test = SyntheticBCDLoader(max(args.batch_size, args.synthetic_pairs // 8), args.batch_size, args.in_height, seed=6)
    return train, val, test, len(train)


@torch.no_grad()
def val(args, val_loader, model, epoch):
    model.eval()
    meter = ConfuseMatrixMeter(n_class=2)
    losses = []
    for img, target in val_loader:
        pre, post = img[:, 0:3].cuda().float(), img[:, 3:6].cuda().float()
        target = target.cuda().float()
        output = model.update_bcd(pre, post)
        losses.append(BCEDiceLoss(output, target))
        meter.update_cm_device(output, target)
    return float(torch.stack(losses).mean()), meter.get_scores()


def train(args, train_loader, model, optimizer, sync, epoch, max_batches, cur_iter=0, lr_factor=1.0):
    model.train()
    meter = ConfuseMatrixMeter(n_class=2)
    losses = []
    lr = args.lr
    for iter_idx, (img, target) in enumerate(train_loader):
        if iter_idx + cur_iter == 3:   # once per run, after the first iterations have built everything long-lived (hostopt.py)
            freeze_gc()
        pre, post = img[:, 0:3].cuda().float(), img[:, 3:6].cuda().float()
        target = target.cuda().float()
        start = time.time()
        lr = adjust_learning_rate(args, optimizer, epoch, iter_idx + cur_iter, max_batches, lr_factor=lr_factor)
        output = model.update_bcd(pre, post)
        loss = BCEDiceLoss(output, target)
        optimizer.zero_grad()
        loss.backward()
        sync.finish()
        optimizer.step()
        losses.append(loss.detach())
        meter.update_cm_device(output, target)
        if (iter_idx + 1) % 5 == 0 and args.rank == 0:
            taken = time.time() - start
            res = (max_batches * args.max_epochs - iter_idx - cur_iter) * taken / 3600
            print(f"[epoch {epoch}] [iter {iter_idx + 1}/{len(train_loader)} {res:.2f}h] "
                  f"[lr {optimizer.param_groups[0]['lr']:.6f}] [bn_loss {loss.item():.4f}]")
    return float(torch.stack(losses).mean()), meter.get_scores(), lr


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
    save_path = osp(args.save_dir, f"{args.dataset}_iter_{args.max_steps}_lr_{args.lr}")
    os.makedirs(save_path, exist_ok=True)
    train_loader, _, test_loader, max_batches = create_data_loaders(args, args.rank)
    args.max_epochs = int(np.ceil(args.max_steps / max_batches))
    start_epoch, cur_iter = 0, 0
    ckpt = osp(save_path, "checkpoint.pth.tar")
    if args.resume is not None and os.path.isfile(ckpt):  # weights + epoch only (reference model/utils.py:205-232)
        state = torch.load(ckpt, map_location="cpu")
        start_epoch = state["epoch"]
        cur_iter = start_epoch * max_batches
        model.load_state_dict(state["state_dict"])
    arena, sync = setup_data_parallel(model, torch.device("cuda", local))
    optimizer = FusedAdam(arena, args.lr, (0.9, 0.99), eps=1e-08, weight_decay=1e-4)
    logger = open(osp(save_path, args.log_file), "a+") if args.rank == 0 else None
    max_F1_val, model_file_name = 0, osp(save_path, "best_model.pth")
    for epoch in range(start_epoch, args.max_epochs):
        loss_train, score_tr, lr = train(args, train_loader, model, optimizer, sync, epoch, max_batches, cur_iter)
        cur_iter += len(train_loader)
        if epoch == 0:
            continue
        # rank 0 validates (BatchNorm running statistics are per rank by design; the checkpoint holds rank 0's, so
        # its score is the one that describes the saved model); the other ranks wait for it on the host (gloo barrier
        # with a long timeout): neither an RCCL collective left pending across a long validation pass nor a process
        # group torn down while rank 0 still writes its checkpoint
        if args.rank != 0:
            host_barrier()
            continue
        loss_val, score_val = val(args, test_loader, model, epoch)
        logger.write("\n%d\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f" % (
            epoch, score_val["Kappa"], score_val["IoU"], score_val["F1"], score_val["recall"], score_val["precision"]))
        logger.flush()
        torch.save({"epoch": epoch + 1, "arch": str(model), "state_dict": model.state_dict(),
                    "optimizer": optimizer.state_dict(), "loss_train": loss_train, "loss_val": loss_val,
                    "F_train": score_tr["F1"], "F_val": score_val["F1"], "lr": lr}, ckpt)
        if max_F1_val <= score_val["F1"]:
            max_F1_val = score_val["F1"]
            torch.save(model.state_dict(), model_file_name)
        print(f"\nEpoch No. {epoch}:\tTrain Loss = {loss_train:.4f}\tVal Loss = {loss_val:.4f}\t"
              f"F1(tr) = {score_tr['F1']:.4f}\tF1(val) = {score_val['F1']:.4f}")
        host_barrier()
    if logger:
        logger.close()
    if world > 1:
        dist.destroy_process_group()


def build_parser():
    p = ArgumentParser()
    p.add_argument("--dataset", default="SYNTH-CD", help="any name containing 'CD' selects the BCD head")
    p.add_argument("--file_root", default="", help="unused (file datasets are out of scope)")
    p.add_argument("--in_height", type=int, default=256)
    p.add_argument("--in_width", type=int, default=256)
    p.add_argument("--num_perception_frame", type=int, default=1)
    p.add_argument("--num_class", type=int, default=1)
    p.add_argument("--max_steps", type=int, default=80000)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--lr", type=float, default=2e-4)
    p.add_argument("--lr_mode", default="poly")
    p.add_argument("--step_loss", type=int, default=100)
    p.add_argument("--pretrained", default="./pretrained/X3D_L.pyth")
    p.add_argument("--save_dir", default="./exp")
    p.add_argument("--resume", default=None)
    p.add_argument("--log_file", default="train_val_log.txt")
    p.add_argument("--gpu_id", default=0, type=int)
    p.add_argument("--synthetic_pairs", type=int, default=256, help="pairs per synthetic epoch")
    p.add_argument("--act_dtype", choices=["f32", "bf16"], default="bf16")
    return p


if __name__ == "__main__":
    trainValidate(build_parser().parse_args())
