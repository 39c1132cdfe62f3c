"""Round-3 GPU tests: SCD validation metrics on the device (c3d_hist2d), the SCD / CC device input pipelines, the SCD
`val()` mirror, world_size-2 data-parallel runs of the SCD and CC paths through the real backward hooks, and the
self-launching `bench.py --gpus 2`.  Reference: scripts/train_SCD.py:104-178, model/utils.py:313-378,
data/transforms.py:300-357, data/dataset.py:411-424, scripts/train_CC.py:466-469."""
import contextlib
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def test_scd_histogram_and_scores_on_device_match_the_reference_fixture(golden_dir):
    """c3d_hist2d + the host arithmetic of SCDD_eval_all against tests/golden/scd_metrics.npz (produced by the REAL
    reference): identical histogram, identical (Fscd, mIoU, SeK)."""
    _need_gpu()
    from change3d_amd.model import utils as mu
    G = np.load(os.path.join(golden_dir, "scd_metrics.npz"))
    nc = int(G["num_class"])
    preds = [torch.from_numpy(p.astype(np.int64)).to(DEV) for p in G["preds"]]
    labels = [torch.from_numpy(l.astype(np.int64)).to(DEV) for l in G["labels"]]
    h = mu.SCDHistogram(nc, torch.device(DEV))
    for p, l in zip(preds, labels):
        h.update(p, l)
    assert np.array_equal(h.matrix(), G["hist"])
    assert tuple(h.scores()) == tuple(G["scores"].tolist())
    assert tuple(mu.SCDD_eval_all(preds, labels, nc)) == tuple(G["scores"].tolist())
    acc = [mu.accuracy(p, l)[0] for p, l in zip(preds, labels)]
    assert np.array_equal(np.array(acc), G["acc"])
    # predictions outside [0, n) are dropped (fast_hist's mask); labels outside raise (numpy's reshape would)
    big = torch.randint(0, nc, (5000,), device=DEV)
    h2 = mu.SCDHistogram(nc, torch.device(DEV))
    h2.update(torch.where(big == 3, torch.full_like(big, 99), big), big)
    assert h2.matrix().sum() == int((big != 3).sum())
    h3 = mu.SCDHistogram(nc, torch.device(DEV))
    h3.update(big, torch.full_like(big, nc))
    with pytest.raises(ValueError):
        h3.matrix()


@pytest.mark.parametrize("shape", [(3, 24, 36), (2, 64, 64)])
def test_scd_device_input_pipeline_is_bit_exact(shape):
    _need_gpu()
    from oracle import transforms as ot
    from change3d_amd.data.transforms import DeviceSCDBatchTransform, SCDTransforms
    B, H, W = shape
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(B, H, W, 6), dtype=np.uint8)
    lab = rng.integers(0, 7, size=(B, H, W, 3), dtype=np.uint8)
    flags = np.array([[b & 1, (b >> 1) & 1, (b + 1) & 1] for b in range(B)], dtype=np.uint8)
    tf = DeviceSCDBatchTransform(DEV)
    pre, post, labels = tf(img, lab, flags)
    torch.cuda.synchronize()
    for b in range(B):
        i_ref, l_ref = ot.scd_transform_sample(img[b], lab[b], flags[b], SCDTransforms.DEFAULT_MEAN, SCDTransforms.DEFAULT_STD)
        assert np.array_equal(pre[b].cpu().numpy(), i_ref[:3]) and np.array_equal(post[b].cpu().numpy(), i_ref[3:])
        assert np.array_equal(labels[b].cpu().numpy(), l_ref)
    assert labels.dtype == torch.int64
    pre0, post0, labels0 = tf(img, lab, None)      # validation transform: no augmentation
    i_ref, l_ref = ot.scd_transform_sample(img[0], lab[0], (0, 0, 0), SCDTransforms.DEFAULT_MEAN, SCDTransforms.DEFAULT_STD)
    assert np.array_equal(pre0[0].cpu().numpy(), i_ref[:3]) and np.array_equal(labels0[0].cpu().numpy(), l_ref)


def test_cc_device_input_pipeline_is_bit_exact():
    _need_gpu()
    from oracle import transforms as ot
    from change3d_amd.data.transforms import DeviceCCBatchTransform
    B, H, W = 5, 32, 40
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, size=(B, 2, 3, H, W), dtype=np.uint8)
    img[0, 0, 0].reshape(-1)[:256] = np.arange(256, dtype=np.uint8)      # every byte value appears
    swap = np.array([0, 1, 0, 1, 1], dtype=np.uint8)
    tf = DeviceCCBatchTransform(DEV)
    pre, post = tf(img, swap)
    torch.cuda.synchronize()
    for b in range(B):
        ref = ot.cc_transform_sample(img[b], bool(swap[b]))
        assert np.array_equal(pre[b].cpu().numpy(), ref[0]) and np.array_equal(post[b].cpu().numpy(), ref[1]), b
    pre2, post2 = tf(img, None)
    assert np.array_equal(pre2[1].cpu().numpy(), ot.cc_transform_sample(img[1], False)[0])


def test_scd_val_mirror_runs_and_matches_a_host_recomputation():
    """`val()` of the SCD script mirror (device histogram, one read-back) against the reference's own bookkeeping
    restated on the host from the same masks: Fscd / mIoU / SeK and the accuracy meter agree exactly."""
    _need_gpu()
    from types import SimpleNamespace
    from oracle import metrics as om_
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.scripts import train_SCD as ts
    args = synth.make_args(num_perception_frame=3, size=64, dataset="SECOND", num_class=7)
    args.act_dtype = torch.float32
    with contextlib.redirect_stdout(io.StringIO()):
        net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
    net = net.to(DEV)
    loader = list(ts.SyntheticSCDLoader(4, 2, 64, 7, seed=3))
    vargs = SimpleNamespace(num_class=7, rank=1)
    Fscd, iou, sek, acc_meter, val_loss = ts.val(vargs, loader, net)
    # host recomputation, the reference way (scripts/train_SCD.py:118-166)
    net.eval()
    preds, labels, accs = [], [], []
    with torch.no_grad():
        for imgs, lab in loader:
            pm, qm, cm = net.update_scd(imgs[:, 0:3].to(DEV).float(), imgs[:, 3:6].to(DEV).float())
            lc = lab[:, 2].long()
            pl, ql = (lab[:, 0].long() * lc).numpy(), (lab[:, 1].long() * lc).numpy()
            chg = (cm.cpu() > 0.5).squeeze(1).long()
            pa, pb = (pm.cpu().argmax(1) * chg).numpy(), (qm.cpu().argmax(1) * chg).numpy()
            for a, b, la, lb in zip(pa, pb, pl, ql):
                accs.append((om_.accuracy(a, la)[0] + om_.accuracy(b, lb)[0]) * 0.5)
                preds += [a, b]
                labels += [la, lb]
    ref = om_.SCDD_eval_all(preds, labels, 7)
    assert (Fscd, iou, sek) == tuple(ref), ((Fscd, iou, sek), ref)
    assert abs(acc_meter.average() - float(np.mean(accs))) < 1e-6
    assert np.isfinite(val_loss.average())


# ------------------------------------------------------------------------------------------- world = 2 on one GPU
SIZE, BATCH = 64, 2


def _scd_local(rank, dev, setup):
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import ChangeSimilarity, CrossEntropyLoss2d, ParamArena
    from change3d_amd.parallel import ordered_hot_params, setup_data_parallel
    from change3d_amd.scripts.train_SCD import scd_loss
    args = synth.make_args(num_perception_frame=3, size=SIZE, dataset="SECOND", num_class=7)
    args.act_dtype = torch.bfloat16
    with contextlib.redirect_stdout(io.StringIO()):
        net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
    net = net.to(dev).train()
    if setup:
        arena, sync = setup_data_parallel(net, dev, overlap=True)
    else:
        arena, sync = ParamArena(ordered_hot_params(net)[0], dev), None
    pre, post, _ = (t.to(dev) for t in synth.synth_batch(BATCH, SIZE, seed=rank))
    labels = synth.synth_scd_labels(BATCH, SIZE, seed=rank).to(dev)
    arena.zero_grad()
    scd_loss(CrossEntropyLoss2d(ignore_index=0), ChangeSimilarity(), net.update_scd(pre, post), labels)[0].backward()
    return [arena], sync, net


def _cc_local(rank, dev, setup):
    from change3d_amd import synthetic as synth
    from change3d_amd.model.caption_decoder import packed_cross_entropy
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.parallel import setup_data_parallel_cc
    args = synth.make_cc_args(size=SIZE, vocab_size=101, dropout=0.0)
    args.act_dtype = torch.bfloat16
    with contextlib.redirect_stdout(io.StringIO()):
        net = Trainer(args)
    sd = synth.synth_state_dict(net, seed=16)
    sd["decoder.position_encoding.pe"] = net.state_dict()["decoder.position_encoding.pe"].clone()
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.decoder.position_encoding.dropout.p = 0.0
    # (without a process group GradSync.world is 1: the same arenas, no exchange -- the single-rank reference run)
    (ea, _), (da, _), both = setup_data_parallel_cc(net, dev, overlap=True)
    pre, post, _ = (t.to(dev) for t in synth.synth_batch(BATCH, SIZE, seed=rank))
    caps, caplens = (t.to(dev) for t in synth.synth_captions(BATCH, seed=rank, vocab_size=101))
    ea.zero_grad(); da.zero_grad()
    feat = net.update_cc(pre, post)
    Bc, Cc, Hc, Wc = feat.shape
    logits = net.decoder.logits_seq_first(feat.permute(2, 3, 0, 1).reshape(Hc * Wc, Bc, Cc), caps)
    packed_cross_entropy(logits, caps, caplens, 101, ignore_index=0).backward()
    return [ea, da], both, net


def _dp_worker(task, rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    arenas, sync, net = (_scd_local if task == "scd" else _cc_local)(rank, dev, True)
    assert sync.world == world
    hooked = net.encoder.x3d.blocks[3 if task == "scd" else 4].post_backward
    assert hooked is not None
    fired = sync._tail_launched if task == "scd" else all(s._tail_launched for s in sync.syncs)
    assert fired, "the stage hook did not launch the overlapped bucket(s) inside backward()"
    sync.finish()
    torch.cuda.synchronize()
    q.put((rank, [a.flat_grad.cpu().numpy() for a in arenas]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("task", ["scd", "cc"])
def test_world2_scd_and_cc_allreduce_equals_mean_of_single_rank_grads(task):
    """SURVEY.md 8(e) for BASELINE configs[3] / [4]: two ranks on one MI355X (gloo carries the exchange) run the REAL
    chain -- backward -> stage hook (SCD: end of res4's backward; CC: end of res5's, which launches the decoder buffer
    AND the res5 tail of the encoder buffer) -> finish() -- and the reduced buffers equal the mean of the two
    single-rank gradient buffers."""
    _need_gpu()
    import torch.multiprocessing as mp
    dev = torch.device("cuda", 0)
    local = _scd_local if task == "scd" else _cc_local
    singles, meta = [], None
    for r in range(2):
        arenas, _, _ = local(r, dev, False)
        torch.cuda.synchronize()
        singles.append([a.flat_grad.cpu().numpy().astype(np.float64) for a in arenas])
        meta = [([int(o) for o in a.offsets], [int(p.numel()) for p in a.params]) for a in arenas]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 2000) + (0 if task == "scd" else 7)
    procs = [ctx.Process(target=_dp_worker, args=(task, r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, f0), (_, f1) = res
    worst = 0.0
    for k, (offs, sizes) in enumerate(meta):
        assert np.array_equal(f0[k], f1[k]), "ranks disagree after the all-reduce"
        mean = 0.5 * (singles[0][k] + singles[1][k])
        for o, n in zip(offs, sizes):
            a, b = f0[k][o:o + n].astype(np.float64), mean[o:o + n]
            worst = max(worst, float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)))
    print(f"world=2 {task}: worst per-parameter rel-L2 vs mean of single-rank gradients {worst:.2e}")
    assert worst < 1e-4, worst


@pytest.mark.parametrize("task", ["bcd", "cc"])
def test_bench_self_launch_two_ranks_on_one_gpu(task):
    """`python bench.py --gpus 2` re-executes itself under torch.distributed.run (one rank per GPU on a real node); here
    both ranks share the box's single GPU and gloo carries the exchange (C3D_DIST_BACKEND / C3D_DIST_DEVICE test hooks):
    the printed line must describe a 2-rank job."""
    _need_gpu()
    env = dict(os.environ, C3D_DIST_BACKEND="gloo", C3D_DIST_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "64",
           "--batch", "2", "--task", task, "--no-cpu-baseline", "--no-kernel-profile", "--no-also"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["dist_world_size"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["config"]["global_batch"] == 4 and d["value"] > 0 and np.isfinite(d["config"]["final_loss"])
