"""Root-cause tool for the deferred decoder weight gradients (engine.DEFER_DECODER_WGRAD).

  part 1 (eager): the two-stream schedule with the streams pushed apart by spin kernels (side stream late / main stream late),
                  gradients against the one-stream schedule;
  part 2 (graph): the captured step with deferral on and off: gradients against eager, and both graphs dumped as DOT
                  (hipGraphDebugDotPrint) into gpurun_out/ for an edge diff (tools/dot_edges.py).

    python tools/dbg_defer.py [eager] [graph]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vinet_amd import _lib, engine as E, loss as VL, model as VM, optim as VO, synth

_lib.load()
E.set_default_dtype("bf16")
DEV = torch.device("cuda:0")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(OUT, exist_ok=True)
B, T, H, W = 2, 8, 64, 96
x = synth.clip(B, T, H, W, 3).permute(0, 2, 1, 3, 4).to(DEV).contiguous()
gt = synth.gt_map(B, H, W, 3).to(DEV)


def fresh():
    m = VM.VideoSaliencyModel(num_clips=T)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 3))
    m = m.to(DEV).train()
    opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    return m, opt


def grads_eager(side, defer, spin_side=0, spin_main=0, passes=2):
    E.WGRAD_SIDE_STREAM, E.DEFER_DECODER_WGRAD, E.DBG_SPIN_SIDE, E.DBG_SPIN_MAIN = side, defer, spin_side, spin_main
    m, opt = fresh()
    for _ in range(passes):
        opt.zero_grad()
        VL.kldiv(m(x), gt).backward()
    torch.cuda.synchronize()
    E.DBG_SPIN_SIDE = E.DBG_SPIN_MAIN = 0
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    return opt.flat_g.clone(), names, opt._offs, [p.numel() for p in opt._params]


def report(tag, g, ref, names, offs, nums, show=8):
    rel = float((g - ref).norm() / ref.norm())
    bad = []
    for n, o, k in zip(names, offs, nums):
        d = float((g[o:o + k] - ref[o:o + k]).norm() / (ref[o:o + k].norm() + 1e-30))
        if d > 1e-4:
            bad.append((d, n))
    print("%-44s rel %.3e   %d / %d parameters off by > 1e-4" % (tag, rel, len(bad), len(names)), flush=True)
    for d, n in sorted(bad, reverse=True)[:show]:
        print("      %.3e %s" % (d, n))
    return rel


what = sys.argv[1:] or ["eager", "graph"]
ref, names, offs, nums = grads_eager(False, 0)
if "eager" in what:
    print("---- eager: one-stream schedule is the reference ----")
    for tag, kw in [("two streams, in order", dict(side=True, defer=0)),
                    ("two streams, deferred", dict(side=True, defer=1)),
                    ("deferred + side stream 0.2 ms late/job", dict(side=True, defer=1, spin_side=400000)),
                    ("deferred + side stream 2 ms late/job", dict(side=True, defer=1, spin_side=4000000)),
                    ("deferred + main stream 0.2 ms late/dgrad", dict(side=True, defer=1, spin_main=400000)),
                    ("deferred + main stream 1 ms late/dgrad", dict(side=True, defer=1, spin_main=2000000)),
                    ("in order + side stream 2 ms late/job", dict(side=True, defer=0, spin_side=4000000)),
                    ("in order + main stream 1 ms late/dgrad", dict(side=True, defer=0, spin_main=2000000))]:
        g, *_ = grads_eager(**kw)
        report(tag, g, ref, names, offs, nums)

if "graph" in what:
    import ctypes
    from vinet_amd.graph import GraphedTrainStep
    hip = ctypes.CDLL("libamdhip64.so")
    print("---- captured step (gradients of the first replay against eager) ----")
    variants = [("tape order", "0", "per_job"), ("deferred, one join per job", "1", "per_job"),
                ("deferred, ONE join for the batch", "1", "once"), ("deferred, a main-stream node between joins", "1", "dummy")]
    for tag, defer_in_capture, mode in variants:
        os.environ["VINET_DBG_DEFER_IN_CAPTURE"] = defer_in_capture
        os.environ["VINET_DBG_FLUSH_MODE"] = mode
        E.WGRAD_SIDE_STREAM, E.DEFER_DECODER_WGRAD = True, 1
        m, opt = fresh()
        E.LAUNCH_LOG = None
        step = GraphedTrainStep(m, opt, VL.kldiv, (x,), gt, keep_graph=True, launch_log=True)
        log = step.launch_log
        step((x,), gt)
        torch.cuda.synchronize()
        report("graph: " + tag, opt.flat_g.clone(), ref, names, offs, nums, show=4)
        key = "defer%s_%s" % (defer_in_capture, mode)
        main = torch.cuda.current_stream().cuda_stream
        with open(os.path.join(OUT, "launches_%s.txt" % key), "w") as f:
            for name, st, tg in log:
                f.write("%s %s %s\n" % ("main" if st == step.capture_stream else "side", name, tg or ""))
        try:
            h = step.graph.raw_cuda_graph()
            path = os.path.join(OUT, "graph_%s.dot" % key).encode()
            rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(h), path, ctypes.c_uint(0))
            n = ctypes.c_size_t(0)
            hip.hipGraphGetNodes(ctypes.c_void_p(h), None, ctypes.byref(n))
            ne = ctypes.c_size_t(0)
            hip.hipGraphGetEdges(ctypes.c_void_p(h), None, None, ctypes.byref(ne))
            print("      dot rc %d, %d nodes, %d edges, %d engine launches logged" % (rc, n.value, ne.value, len(log)), flush=True)
        except Exception as e:
            print("      graph dump failed:", e)
    os.environ.pop("VINET_DBG_DEFER_IN_CAPTURE", None)
    os.environ.pop("VINET_DBG_FLUSH_MODE", None)
