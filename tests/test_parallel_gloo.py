"""Data-parallel path (vinet_amd/parallel.py) with world_size 2 on CPU / gloo.

Two processes each run the engine (C ABI served by the CPU emulator) on their shard
of the batch; after the flat-buffer all-reduce + fused Adam with grad_scale = 1/world
both replicas must hold the parameters a single process gets on the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed=5):
    from vinet_amd import model as VM
    from vinet_amd import synth
    m = VM.DecoderConvUp8()          # no BatchNorm: shard gradients average to the full-batch gradient exactly
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed))
    return m


def _inputs(B):
    from vinet_amd import synth
    shapes = [(B, 1024, 1, 1, 2), (B, 832, 2, 2, 4), (B, 480, 4, 4, 8), (B, 192, 4, 8, 16)]
    ys = [synth.normal("ddp_y%d" % i, s, 9).abs() for i, s in enumerate(shapes)]
    gt = synth.gt_map(B, 32, 64, 9)
    return ys, gt


def _one_step(m, ys, gt, world_hook=None):
    from vinet_amd import loss as VL
    from vinet_amd import optim as VO
    from vinet_amd import parallel
    opt = VO.Adam(m.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt)
    opt.zero_grad()
    l = VL.kldiv(m(*ys), gt)
    l.backward()
    parallel.allreduce_gradients(opt)
    opt.step()
    return opt, float(l)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), VINET_RDZV_FILE=os.path.join(out, "rdzv"))   # file rendezvous: no port
    torch.set_num_threads(2)
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import engine as E
    from vinet_amd import parallel
    L._install_test_double(AbiEmulator())
    E.set_default_dtype("fp32")
    r, w, _, dev = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    m = _build(seed=5 + rank)        # deliberately different: broadcast_parameters must fix it
    ys, gt = _inputs(4)
    lo, hi = parallel.shard_batch(4, rank, world)
    opt, l = _one_step(m, [y[lo:hi] for y in ys], gt[lo:hi])
    torch.save(dict(p=opt.flat_p.clone(), loss=l), os.path.join(out, "rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _worker_graphed(rank, world, port, out):
    """the captured-step wrapper under data parallelism: step = [zero_grad, forward, loss, backward] (a replayed hipGraph on a GPU,
    the same calls eagerly on the ABI emulator) -> ONE all-reduce of the flat gradient buffer -> fused Adam with 1 / world"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), VINET_RDZV_FILE=os.path.join(out, "rdzv"))   # file rendezvous: no port
    torch.set_num_threads(2)
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import engine as E
    from vinet_amd import loss as VL
    from vinet_amd import optim as VO
    from vinet_amd import parallel
    from vinet_amd.graph import GraphedTrainStep
    L._install_test_double(AbiEmulator())
    E.set_default_dtype("fp32")
    parallel.init_from_env(backend="gloo")
    m = _build(seed=5 + rank)
    ys, gt = _inputs(4)
    lo, hi = parallel.shard_batch(4, rank, world)
    opt = VO.Adam(m.parameters(), lr=1e-3)
    parallel.broadcast_parameters(opt)
    mine, gmine = tuple(y[lo:hi].contiguous() for y in ys), gt[lo:hi].contiguous()
    step = GraphedTrainStep(m, opt, VL.kldiv, mine, gmine)
    l = float(step(mine, gmine))
    assert opt.grad_scale == 0.5
    torch.save(dict(p=opt.flat_p.clone(), loss=l), os.path.join(out, "rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("worker", ["eager", "graphed_step"])
def test_two_ranks_match_single_process(tmp_path, worker):
    port = _free_port()
    mp.spawn(_worker if worker == "eager" else _worker_graphed, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0["p"], r1["p"]), "replicas diverged"
    # single process, whole batch
    sys.path.insert(0, ROOT)
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import engine as E
    L._install_test_double(AbiEmulator())
    E.set_default_dtype("fp32")
    try:
        m = _build(seed=5)
        ys, gt = _inputs(4)
        opt, l = _one_step(m, ys, gt)
    finally:
        L._install_test_double(None)
        E.set_default_dtype("bf16")
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - l) < 1e-5
    # Adam's first step is lr * sign(g) wherever |g| >> eps; compare the updates
    d = (r0["p"] - opt.flat_p).abs()
    frac_off = float((d > 2e-4).float().mean())
    assert frac_off < 2e-3, "%.4f of the parameters moved differently" % frac_off


def test_shard_batch():
    from vinet_amd import parallel
    assert parallel.shard_batch(8, 0, 2) == (0, 4) and parallel.shard_batch(8, 1, 2) == (4, 8)
    with pytest.raises(AssertionError):
        parallel.shard_batch(7, 0, 2)


# ---- ViNet-8 with BatchNorm: bucketed overlapped all-reduce == one flat all-reduce == torch DDP -------------------------------
def _vinet8(seed):
    from vinet_amd import model as VM
    from vinet_amd import synth
    m = VM.VideoSaliencyModel(num_clips=8)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed))
    return m.train()


def _vinet_batch(B):
    from vinet_amd import synth
    return synth.clip(B, 8, 32, 32, 3).permute(0, 2, 1, 3, 4), synth.gt_map(B, 32, 32, 3)


def _worker_vinet(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), VINET_RDZV_FILE=os.path.join(out, "rdzv"))   # file rendezvous: no port
    torch.set_num_threads(2)
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import engine as E
    from vinet_amd import loss as VL
    from vinet_amd import optim as VO
    from vinet_amd import parallel
    L._install_test_double(AbiEmulator())
    E.set_default_dtype("fp32")
    parallel.init_from_env(backend="gloo")
    x, gt = _vinet_batch(world)
    xs, gs = x[rank:rank + 1], gt[rank:rank + 1]
    res = {}
    # (a) one flat all-reduce after backward
    m = _vinet8(7)
    opt = VO.Adam(parallel.trainable_parameters(m), lr=1e-3)
    opt.zero_grad()
    VL.kldiv(m(xs), gs).backward()
    parallel.allreduce_gradients(opt)
    res["flat"] = opt.flat_g.clone() * opt.grad_scale
    res["bn_rm"] = m.backbone.base1[0].bn_s.running_mean.clone()
    # (b) bucketed, issued from the tape (tiny buckets: many of them, every split point exercised)
    m = _vinet8(7)
    opt = VO.Adam(parallel.trainable_parameters(m), lr=1e-3)
    bk = parallel.GradientBuckets(opt, bucket_bytes=1 << 20)
    assert len(bk.buckets) > 8 and bk.buckets[0][1] == opt.flat_g.numel() and bk.buckets[-1][0] == 0
    opt.zero_grad()
    bk.begin_step()
    VL.kldiv(m(xs), gs).backward()
    launched_in_backward = sum(bk._launched)
    bk.finish()
    res["bucketed"] = opt.flat_g.clone() * opt.grad_scale
    res["launched_in_backward"] = launched_in_backward
    res["nbuckets"] = len(bk.buckets)
    # (c) torch DistributedDataParallel around the same module, torch.optim-style access to .grad
    m = _vinet8(7)
    ddp = parallel.ddp_wrap(m)
    assert E.param_grad_mode() == "autograd"
    VL.kldiv(ddp(xs), gs).backward()
    res["ddp"] = torch.cat([p.grad.reshape(-1) for p in parallel.trainable_parameters(m)])
    res["order"] = [p.numel() for p in parallel.trainable_parameters(m)]
    # (d) torch.autograd.grad returns real parameter gradients in that mode
    m2 = _vinet8(7)
    ps = parallel.trainable_parameters(m2)[:4]
    gl = torch.autograd.grad(VL.kldiv(m2(xs), gs), ps)
    res["autograd_grad_ok"] = all(g is not None and g.shape == p.shape and float(g.abs().sum()) > 0 for g, p in zip(gl, ps)) and all(p.grad is None for p in ps)
    E.set_param_grad_mode("fused")
    torch.save(res, os.path.join(out, "vinet_rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("world", [2, 4])
def test_vinet8_bucketed_allreduce_and_ddp(tmp_path, world):
    """ViNet-8 (BatchNorm in training mode, per-replica statistics) on `world` gloo ranks: the bucketed all-reduce the tape
    issues during backward, the one-shot flat all-reduce and torch's DistributedDataParallel around the same module
    all produce the same averaged gradient; replicas agree; BatchNorm statistics stay per replica."""
    port = _free_port()
    mp.spawn(_worker_vinet, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, "vinet_rank%d.pt" % r)) for r in range(world)]
    r0 = rs[0]
    assert r0["nbuckets"] > 8 and r0["launched_in_backward"] >= r0["nbuckets"] - 1, "buckets must go out from inside backward"
    assert r0["autograd_grad_ok"]
    for r in rs[1:]:
        assert torch.equal(r["flat"], r0["flat"]) and torch.equal(r["bucketed"], r0["bucketed"])
        assert not torch.equal(r["bn_rm"], r0["bn_rm"]), "BatchNorm statistics are per replica (train.py:181-185 semantics)"
    # (with more than two ranks the ring's summation order depends on the chunking of the tensor: equal to fp32 round-off)
    if world == 2:
        assert torch.equal(r0["bucketed"], r0["flat"]), "bucketing must not change the reduction"
    db = (r0["bucketed"] - r0["flat"]).abs().max() / (r0["flat"].abs().max() + 1e-30)
    assert float(db) < 1e-6, "bucketed and flat all-reduce disagree: %g" % float(db)
    # flat buffer slots are 16-byte aligned: gather the parameter ranges before comparing with DDP's .grad
    off, chunks = 0, []
    for n in r0["order"]:
        chunks.append(r0["flat"][off:off + n])
        off += (n + 3) // 4 * 4
    flat = torch.cat(chunks)
    d = (flat - r0["ddp"]).abs().max() / (flat.abs().max() + 1e-30)
    assert float(d) < 1e-5, "DDP and the flat all-reduce disagree: %g" % float(d)


def test_trainable_parameters_skip_soundnet_heads():
    """conv8_objs / conv8_scns (11.48 M parameters, model.py:788-791) never get a gradient: not in the buffer"""
    from vinet_amd import model as VM
    from vinet_amd import parallel
    m = VM.VideoAudioSaliencyModel(num_clips=32)
    allp = sum(p.numel() for p in m.parameters())
    used = sum(p.numel() for p in parallel.trainable_parameters(m))
    assert allp - used == 1024 * 1000 * 8 + 1000 + 1024 * 401 * 8 + 401


def test_gradient_buckets_count_each_parameter_once(monkeypatch):
    """A bucket leaves when every member PARAMETER has reported, not after as many reports as it has members: a parameter
    that reports twice (a module used twice in forward) must not release the bucket early, and a report behind the
    bucket's all-reduce is an error (it would add an un-reduced contribution to a summed slice)."""
    import types
    from vinet_amd import parallel as P
    ps = [torch.nn.Parameter(torch.zeros(4)) for _ in range(4)]
    opt = types.SimpleNamespace(_params=ps, _offs=[0, 4, 8, 12], flat_g=torch.zeros(16), grad_scale=1.0)
    gb = P.GradientBuckets(opt, bucket_bytes=32)        # two parameters per bucket, last parameters first
    assert [(lo, hi, n) for lo, hi, n in gb.buckets] == [(8, 16, 2), (0, 8, 2)]
    sent = []
    monkeypatch.setattr(P, "distributed", lambda: True)
    monkeypatch.setattr(P.dist, "get_world_size", lambda: 2)
    monkeypatch.setattr(gb, "_launch", lambda b, ctx=None: (sent.append(b), gb._launched.__setitem__(b, True)))
    gb.begin_step()
    gb._on_param(None, ps[3])
    assert sent == []
    gb._on_param(None, ps[2])
    assert sent == [0]
    with pytest.raises(AssertionError):
        gb._on_param(None, ps[3])                       # a second writer of an already reduced slice
    # a parameter with two writers: announce it, the bucket waits for the second report
    gb.expect_reports({id(ps[1]): 2})
    gb.begin_step()
    sent.clear()
    gb._on_param(None, ps[1])
    gb._on_param(None, ps[0])
    assert sent == []
    gb._on_param(None, ps[1])
    assert sent == [1]


def _occupied_port_worker(rank, out, port, set_env):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    for k in ("MASTER_ADDR", "MASTER_PORT", "VINET_RDZV_FILE"):
        os.environ.pop(k, None)
    if set_env:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from vinet_amd import parallel
    parallel.FORCE_COLLECTIVES = True
    parallel.init_from_env(backend="gloo")
    t = torch.ones(3)
    dist.all_reduce(t)
    torch.save(dict(ok=bool((t == 1).all()), port=int(os.environ["MASTER_PORT"])), os.path.join(out, "occ.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["default_29500", "explicit_env_port"])
def test_one_rank_group_survives_an_occupied_rendezvous_port(tmp_path, which):
    """the driver's GPU box had the rendezvous port taken and `init_process_group` died with EADDRINUSE.  A
    one-rank group (bench.py --force-collectives, train.py under FORCE_COLLECTIVES) has nobody to agree a port with, so
    init_from_env moves to a port the kernel hands out -- with the default 29500 AND an explicit MASTER_PORT both held by a
    listening socket of this test."""
    held = socket.socket()
    held.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        if which == "default_29500":
            try:
                held.bind(("", 29500))
            except OSError:
                pass                      # somebody else already holds it: equally occupied
            else:
                held.listen(1)
            port = 29500
        else:
            held.bind(("", 0))
            held.listen(1)
            port = held.getsockname()[1]
        mp.spawn(_occupied_port_worker, args=(str(tmp_path), port, which != "default_29500"), nprocs=1, join=True)
    finally:
        held.close()
    r = torch.load(os.path.join(tmp_path, "occ.pt"))
    assert r["ok"] and r["port"] != port, r


def test_multi_rank_group_on_an_occupied_port_names_the_ways_out(monkeypatch):
    from vinet_amd import parallel
    calls = []

    def boom(*a, **k):
        calls.append(1)
        raise RuntimeError("The server socket has failed to listen on any local network address. port: 29500, useIpv6: false, "
                           "code: -98, name: EADDRINUSE, message: address already in use")

    monkeypatch.setattr(parallel.dist, "init_process_group", boom)
    monkeypatch.setattr(parallel.dist, "is_initialized", lambda: False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("VINET_RDZV_FILE", raising=False)
    with pytest.raises(RuntimeError, match="VINET_RDZV_FILE"):
        parallel.init_from_env(backend="gloo")
    assert len(calls) == 1
