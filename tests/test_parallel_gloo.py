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
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
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


@pytest.mark.slow
def test_two_ranks_match_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
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
