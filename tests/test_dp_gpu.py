"""Data-parallel path on a GPU with world_size 2: two ranks share ONE MI355X (gloo carries the exchange; on an
8-GPU node the backend is nccl = RCCL) and run the REAL chain -- `loss.backward()` -> the stage hook inside
`_StageFn.backward` (side-stream join, then `GradSync.launch_tail` on the comm stream while res3/res2/stem backward
and the side-stream weight gradients are still running) -> `GradSync.finish()`.  The reduced flat gradient buffer
must equal the mean of the two single-rank gradient buffers computed without any process group (SURVEY.md 8(e):
"mean over ranks of local-loss gradients")."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, BATCH = 64, 2


def _local_step(rank, dev, dtype, sync_setup):
    """One forward/backward of rank `rank`'s shard; returns (arena, sync, net)."""
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss, ParamArena
    from change3d_amd.parallel import ordered_hot_params, setup_data_parallel
    args = synth.make_args(size=SIZE)
    args.act_dtype = dtype
    net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
    net = net.to(dev).train()
    if sync_setup:
        arena, sync = setup_data_parallel(net, dev, overlap=True)
    else:
        named, _ = ordered_hot_params(net)
        arena, sync = ParamArena(named, dev), None
    pre, post, tgt = (t.to(dev) for t in synth.synth_batch(BATCH, SIZE, seed=rank))
    arena.zero_grad()
    loss = BCEDiceLoss(net.update_bcd(pre, post), tgt)
    loss.backward()
    return arena, sync, net


def _worker(rank, world, port, dtype_name, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dtype = getattr(torch, dtype_name)
    arena, sync, net = _local_step(rank, dev, dtype, sync_setup=True)
    assert sync.world == world and net.encoder.x3d.blocks[3].post_backward is not None
    assert sync._tail_launched, "the stage hook did not fire inside backward()"
    sync.finish()
    torch.cuda.synchronize()
    q.put((rank, arena.flat_grad.cpu().numpy(), int(sync.split)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_world2_real_backward_hook_allreduce_equals_mean_of_single_rank_grads(dtype_name):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import torch.multiprocessing as mp
    dev = torch.device("cuda", 0)
    singles = []
    for r in range(2):
        arena, _, _ = _local_step(r, dev, getattr(torch, dtype_name), sync_setup=False)
        torch.cuda.synchronize()
        singles.append(arena.flat_grad.cpu().numpy().astype(np.float64))
        offs, sizes = [int(o) for o in arena.offsets], [int(p.numel()) for p in arena.params]
    mean = 0.5 * (singles[0] + singles[1])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, f0, split), (_, f1, _) = res
    assert np.array_equal(f0, f1), "ranks disagree after the all-reduce"
    assert 0 < split < f0.size
    worst = 0.0
    for o, n in zip(offs, sizes):
        a, b = f0[o:o + n].astype(np.float64), mean[o:o + n]
        worst = max(worst, float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)))
    print(f"world=2 ({dtype_name}): worst per-parameter rel-L2 vs mean of single-rank gradients {worst:.2e}; "
          f"overlapped tail bucket = {f0.size - split} of {f0.size} floats")
    # single-rank reruns differ only by f32 leaf-gradient atomics (test_step_is_reproducible: < 2e-5)
    assert worst < 1e-4, worst


def _rccl_worker(port, q):
    """One rank, backend nccl (= RCCL): the group has a single member, but GradSync is told world = 2 so that the
    overlapped tail all-reduce, the head all-reduce and the stream fences actually run through RCCL."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from change3d_amd import synthetic as synth
    from change3d_amd.model.trainer import Trainer
    from change3d_amd.model.utils import BCEDiceLoss
    from change3d_amd.parallel import setup_data_parallel
    args = synth.make_args(size=SIZE)
    args.act_dtype = torch.bfloat16
    net = Trainer(args)
    net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
    net = net.to(dev).train()
    arena, sync = setup_data_parallel(net, dev, overlap=True)
    sync.world = 2                                                    # force the collective path (sum over 1 rank, / 2)
    net.encoder.x3d.blocks[3].post_backward = sync.launch_tail
    pre, post, tgt = (t.to(dev) for t in synth.synth_batch(BATCH, SIZE, seed=0))
    out = []
    for it in range(3):                                               # several steps: the comm stream is reused
        arena.zero_grad()
        BCEDiceLoss(net.update_bcd(pre, post), tgt).backward()
        fired = sync._tail_launched
        sync.finish()
        torch.cuda.synchronize()
        out.append((fired, arena.flat_grad.cpu().numpy().copy()))
    # ---- the end of a data-parallel step, as bench.py / scripts/train_BCD.py issue it: GradSync.finish() (head all-reduce on
    # the compute stream, fence on the communication stream, mul_(1 / world)) and FusedAdam.launch() are ENQUEUED back to back
    # on one stream -- no host synchronisation in between (torch's sync debug mode turns any into an error; an explicit
    # torch.cuda.synchronize() is patched to raise), and the Adam kernel sees the AVERAGED gradients (eps = 1 makes the update
    # proportional to the gradient, so an update computed from the unscaled buffer would be twice as large)
    from change3d_amd.model.utils import FusedAdam
    opt = FusedAdam(arena, lr=1e-2, betas=(0.9, 0.99), eps=1.0, weight_decay=0.0)
    arena.zero_grad()
    BCEDiceLoss(net.update_bcd(pre, post), tgt).backward()
    hp = opt.prepare_step()
    torch.cuda.synchronize()
    p_before = arena.flat_param.clone()
    stream_before = torch.cuda.current_stream().cuda_stream
    real_sync = torch.cuda.synchronize

    def no_sync(*a, **k):
        raise AssertionError("host synchronisation between GradSync.finish() and FusedAdam.launch()")
    torch.cuda.synchronize = no_sync
    torch.cuda.set_sync_debug_mode("error")
    try:
        sync.finish()
        opt.launch(*hp)
        same_stream = torch.cuda.current_stream().cuda_stream == stream_before
    finally:
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize = real_sync
    torch.cuda.synchronize()
    g = arena.flat_grad.double()                       # averaged gradients (Adam does not modify them)
    lr, bc1, bc2s = hp
    m, v = 0.1 * g, 0.01 * g * g
    want = p_before.double() - lr * (m / bc1) / (v.sqrt() / bc2s + 1.0)
    upd = (arena.flat_param.double() - p_before.double())
    adam_err = ((arena.flat_param.double() - want).norm() / (want - p_before.double()).norm()).item()
    q.put((dist.get_backend(), out, dict(same_stream=bool(same_stream), adam_err=adam_err, upd_norm=upd.norm().item())))
    dist.destroy_process_group()


def test_overlapped_allreduce_through_rccl_single_rank():
    """The real backend: `torch.distributed` "nccl" (RCCL) on the GPU.  With one member the sum is the identity, so the
    averaged buffer must be exactly half of the gradient -- what is exercised is RCCL initialisation, the all-reduce of
    the tail bucket on the communication stream from inside backward, the head all-reduce, and the stream fences."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import torch.multiprocessing as mp
    dev = torch.device("cuda", 0)
    arena, _, _ = _local_step(0, dev, torch.bfloat16, sync_setup=False)
    torch.cuda.synchronize()
    ref = arena.flat_grad.cpu().numpy().astype(np.float64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29700 + (os.getpid() % 2000), q))
    p.start()
    backend, out, tail = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0 and backend == "nccl"
    # finish() + launch(): one stream, no host sync, Adam on the averaged buffer (an update from the unscaled gradients
    # would be off by a factor of two: relative error 1.0)
    assert tail["same_stream"] and tail["upd_norm"] > 0 and tail["adam_err"] < 1e-3, tail
    for fired, flat in out:
        assert fired, "the stage hook did not launch the tail bucket"
        err = np.linalg.norm(flat.astype(np.float64) - 0.5 * ref) / np.linalg.norm(0.5 * ref)
        assert err < 1e-4, err


def test_side_join_then_host_readback_and_allreduce():
    """The contract of c3d_side_join (include/change3d_hip.h): the fork / done marks between the caller's stream and the
    library's side stream carry no system-scope fence, so whoever reads the weight gradients from OUTSIDE the device right
    behind the join relies on the caller's own stream-ordered operation.  Two such consumers, with NO device synchronisation
    between `backward()` (whose end-of-pass callback is the join) and the read: (1) an asynchronous device-to-host copy of the
    whole flat gradient buffer enqueued on the compute stream, (2) a sum all-reduce of it through RCCL (one-member group).
    Both must equal the buffer as read after a full device synchronisation, bit for bit, over several steps (the weight
    gradients are the LAST kernels of the pass on the side stream: a missing release shows as stale values here)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_join_readback_worker, args=(29500 + ((os.getpid() + 777) % 2000), q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res == "ok", res


def _join_readback_worker(port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from change3d_amd import synthetic as synth
        from change3d_amd.model.trainer import Trainer
        from change3d_amd.model.utils import BCEDiceLoss, ParamArena
        from change3d_amd.parallel import ordered_hot_params
        args = synth.make_args(size=SIZE)
        args.act_dtype = torch.bfloat16
        net = Trainer(args)
        net.load_state_dict(synth.synth_state_dict(net, seed=16, mask_margin=0.25))
        net = net.to(dev).train()
        named, _ = ordered_hot_params(net)
        arena = ParamArena(named, dev)
        host = torch.empty(arena.flat_grad.numel(), dtype=torch.float32).pin_memory()
        red = torch.empty_like(arena.flat_grad)
        for it in range(4):
            pre, post, tgt = (t.to(dev) for t in synth.synth_batch(BATCH, SIZE, seed=it))
            arena.zero_grad()
            torch.cuda.synchronize()
            BCEDiceLoss(net.update_bcd(pre, post), tgt).backward()      # ends with the side-stream join; nothing else waits
            host.copy_(arena.flat_grad, non_blocking=True)               # (1) D2H right behind the join, same stream
            red.copy_(arena.flat_grad)
            dist.all_reduce(red)                                         # (2) RCCL right behind the join (sum over one rank)
            torch.cuda.current_stream().synchronize()
            got_host = host.clone()
            torch.cuda.synchronize()
            ref = arena.flat_grad.cpu()
            if not torch.equal(got_host, ref):
                q.put(f"step {it}: host read-back differs from the synchronised buffer by {(got_host - ref).abs().max().item():.3e}")
                return
            if not torch.equal(red.cpu(), ref):
                q.put(f"step {it}: all-reduced buffer differs by {(red.cpu() - ref).abs().max().item():.3e}")
                return
            if not (torch.isfinite(ref).all() and ref.abs().max().item() > 0):
                q.put(f"step {it}: empty gradient buffer")
                return
        q.put("ok")
    finally:
        dist.destroy_process_group()
