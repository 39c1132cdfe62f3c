#!/usr/bin/env python
"""`scripts/train_CC.py`-shaped driver for the MI355X-native change-captioning path (reference
scripts/train_CC.py:75-168 `train`, :420-520 `main`): same step (encoder(output_final=True) -> 'b c h w -> (h w) b c'
-> CaptionDecoder -> packed cross-entropy -> zero_grad x2 -> backward -> clip_gradient -> encoder / decoder Adam steps
with StepLR(900, gamma=1)), same hyper-parameters (Adam lr 1e-4, weight_decay 1e-5, grad_clip 5, dropout 0.1, 8 heads,
3 layers, embed_dim 192), and the beam-search captioning of `evaluate()` (:170-330; `--eval_pairs N`).  File datasets,
the word map and the caption metrics (BLEU / METEOR / ROUGE / CIDEr: host-side text scoring, `eval_func/`) are outside
SURVEY.md section 8: `--dataset SYNTH-CC` draws LEVIR-CC-shaped synthetic pairs + token sequences.

    python -m change3d_amd.scripts.train_CC --batch_size 16 --max_steps 20 --act_dtype bf16
"""
import os
import sys
import time
from argparse import ArgumentParser

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from change3d_amd import synthetic as synth  # noqa: E402
from change3d_amd.hostopt import freeze_gc  # noqa: E402
from change3d_amd.model.caption_decoder import packed_cross_entropy  # noqa: E402
from change3d_amd.model.trainer import Trainer  # noqa: E402
from change3d_amd.model.utils import FusedAdam, clip_gradient  # noqa: E402
from change3d_amd.parallel import broadcast_module_state, setup_data_parallel_cc  # noqa: E402


def build(args, device):
    model = Trainer(args).to(device)
    broadcast_module_state(model)
    # one flat buffer per optimizer; under torch.distributed.run both are all-reduced from inside backward (the decoder's
    # and res5's buckets when res5's backward returns: change3d_amd/parallel.py::setup_data_parallel_cc)
    (enc_arena, _), (dec_arena, _), args.grad_sync = setup_data_parallel_cc(model, device)
    # reference scripts/train_CC.py:436-458 (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8)
    enc_opt = FusedAdam(enc_arena, args.encoder_lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    dec_opt = FusedAdam(dec_arena, args.decoder_lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    return model, enc_opt, dec_opt


def train_step(args, model, enc_opt, dec_opt, imgs_a, imgs_b, caps, caplens):
    """reference scripts/train_CC.py:111-150; returns (loss, stats) as DEVICE tensors (no host sync)."""
    feat = model.update_cc(imgs_a, imgs_b)                                  # encoder(imgs_A, imgs_B, output_final=True)
    B, C, H, W = feat.shape
    memory = feat.permute(2, 3, 0, 1).reshape(H * W, B, C)                   # rearrange 'b c h w -> (h w) b c'
    logits = model.decoder.logits_seq_first(memory, caps)
    loss, stats = packed_cross_entropy(logits, caps, caplens, args.vocab_size, ignore_index=0, return_stats=True)
    dec_opt.zero_grad()
    enc_opt.zero_grad()
    loss.backward()
    args.grad_sync.finish()          # data-parallel: mean over ranks of the local gradients (no-op on one rank)
    if args.grad_clip is not None:
        clip_gradient(dec_opt, args.grad_clip)
        clip_gradient(enc_opt, args.grad_clip)
    enc_opt.step()
    dec_opt.step()
    return loss.detach(), stats


@torch.no_grad()
def evaluate(args, model, pairs, start_id, end_id, pad_id=0):
    """reference scripts/train_CC.py:170-399 without the file I/O and the text metrics: eval mode (BatchNorm folded
    into the encoder weights), one pair at a time as the reference's batch_size=1 loader does, beam search of width
    `args.beam_size`; returns the hypotheses (special tokens stripped, :345) -- `None` entries are pairs for which no
    beam emitted <end> (the reference records no caption for them, :326-328)."""
    model.eval()
    hyps = []
    for pre, post in pairs:
        feat = model.update_cc(pre, post)                                       # (1, 192, H/16, W/16)
        B, C, H, W = feat.shape
        memory = feat.permute(2, 3, 0, 1).reshape(H * W, B, C)
        best, _, _ = model.decoder.beam_search(memory, start_id, end_id, args.beam_size)
        hyps.append(None if best is None else [w for w in best if w not in (start_id, end_id, pad_id)])
    return hyps


def main():
    p = ArgumentParser()
    p.add_argument("--dataset", default="SYNTH-CC")
    p.add_argument("--n_head", type=int, default=8)
    p.add_argument("--n_layer", type=int, default=3)
    p.add_argument("--embed_dim", type=int, default=192)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--num_perception_frame", type=int, default=1)
    p.add_argument("--in_height", type=int, default=256)
    p.add_argument("--in_width", type=int, default=256)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--print_freq", type=int, default=100)
    p.add_argument("--encoder_lr", type=float, default=1e-4)
    p.add_argument("--decoder_lr", type=float, default=1e-4)
    p.add_argument("--grad_clip", type=float, default=5.0)
    p.add_argument("--pretrained", default="model/X3D_L.pyth")
    p.add_argument("--vocab_size", type=int, default=501, help="len(WORDMAP) of the reference; synthetic here")
    p.add_argument("--max_steps", type=int, default=100)
    p.add_argument("--beam_size", type=int, default=1, help="reference default (scripts/train_CC.py:600)")
    p.add_argument("--eval_pairs", type=int, default=0, help="caption this many synthetic pairs after training")
    p.add_argument("--act_dtype", choices=["bf16", "f32"], default="bf16")
    args = p.parse_args()
    if "CC" not in args.dataset:
        raise SystemExit("--dataset must name a change-captioning set (contains 'CC')")
    args.act_dtype = torch.bfloat16 if args.act_dtype == "bf16" else torch.float32
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(16)
    model, enc_opt, dec_opt = build(args, device)
    model.train()
    pre, post, _ = (t.to(device) for t in synth.synth_batch(args.batch_size, args.in_height, seed=rank))
    caps, caplens = (t.to(device) for t in synth.synth_captions(args.batch_size, seed=rank, vocab_size=args.vocab_size))
    start = time.time()
    for i in range(args.max_steps):
        if i == 3:   # once per run, after the first iterations have built everything long-lived (hostopt.py)
            freeze_gc()
        loss, stats = train_step(args, model, enc_opt, dec_opt, pre, post, caps, caplens)
        if rank == 0 and (i % args.print_freq == 0 or i == args.max_steps - 1):
            s = stats.cpu()
            print(f"step: {i}/{args.max_steps} Loss: {loss.item():.4f} Top-1 Accuracy: {100.0 * s[2].item() / max(s[1].item(), 1):.4f} "
                  f"Batch_time: {(time.time() - start) / (i + 1):.4f}s")
    if args.eval_pairs and rank == 0:
        n = args.eval_pairs
        ep, eq, _ = (t.to(device) for t in synth.synth_batch(n, args.in_height, seed=1))
        start_id, end_id = args.vocab_size - 2, args.vocab_size - 1            # synthetic word map: <start>, <end> last
        torch.cuda.synchronize()
        t0 = time.time()
        hyps = evaluate(args, model, [(ep[i:i + 1], eq[i:i + 1]) for i in range(n)], start_id, end_id)
        torch.cuda.synchronize()
        done = [h for h in hyps if h is not None]
        print(f"evaluate: {n} pairs, beam {args.beam_size}, {len(done)} captions, mean length "
              f"{sum(map(len, done)) / max(len(done), 1):.1f}, {(time.time() - t0) / n * 1e3:.1f} ms/pair")


if __name__ == "__main__":
    main()
