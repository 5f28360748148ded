#!/usr/bin/env python3
"""ViNet training throughput on MI355X (BASELINE.json metric: clips/sec training,
32x224x384 bf16, 1/2/4/8 GPUs).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = synthetic batch already in HBM -> forward -> kldiv -> backward ->
(RCCL all-reduce of the flat gradient buffer when N > 1) -> fused Adam, all on the
hand-written HIP kernels of libvinet_hip.so.  Weak scaling: the per-GPU batch is
fixed.  Prints ONE JSON line on rank 0 with the roofline of the dominant kernel
site (HIP events on the launch stream, measured inside the timed region) and the
CPU baseline (PyTorch-CPU oracle on the box's host cores, rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA
# algorithmic work per clip (training): (clip, height, width) -> (GFLOP, MB); config 5 is the
# build-defined 64-frame variant (no reference parity)
TRAIN_WORK = {(32, 224, 384): (675.0, 3014.7), (64, 256, 448): (1800.0, 7994.0), (8, 128, 192): (48.1, 219.0), (8, 224, 384): (168.5, 767.0)}
STEP_MB_PER_GPU = 1119.6
SWEEP_BATCHES = (1, 2, 4, 8, 16, 32)
GLOBAL_BATCH_SWEEP = (64, 256, 1536)      # N > 1: global batches of the data-parallel sweep (8 / 32 / 192 clips per GPU at N = 8)
# forward work per clip (inference): (clip, height, width) -> (GFLOP, MB)
INFER_WORK = {(32, 224, 384): (229.32, 1004.9), (64, 256, 448): (611.5, 2664.8), (8, 128, 192): (16.36, 73.0), (8, 224, 384): (57.25, 255.5)}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (weak scaling; the reference default global batch is 8); "
                    "0 = 192 at 32x224x384, 64 at 64x256x448 (what fits 288 GB with margin)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the local-batch sweep {1,2,4,8,16,32} that the N = 1 training line carries (SURVEY.md 8(d))")
    ap.add_argument("--sweep-steps", type=int, default=5)
    ap.add_argument("--model", choices=["vinet", "avinet"], default="vinet",
                    help="avinet = BASELINE config 4: VideoAudioSaliencyModel with the SoundNet branch + bilinear fusion (32x224x384 only)")
    ap.add_argument("--clip", type=int, default=32)
    ap.add_argument("--height", type=int, default=224)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp32s"],
                    help="fp32s = fp32 tensors, convs on three bf16 MFMAs per product (split operands): the fast configuration inside the 1e-3 contract")
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="infer mode: replay a captured hipGraph")
    ap.add_argument("--no-side-stream", action="store_true", help="run wgrad on the main stream (A/B)")
    ap.add_argument("--profile-all", action="store_true", help="print the per-site HIP-event table to stderr")
    ap.add_argument("--main-priority", type=int, default=0, help="tuning: run the step on a stream of this priority (-1 = high) instead of the default stream")
    ap.add_argument("--no-extras", action="store_true", help="skip the `inference` and `other_configs` legs the default N = 1 training line carries")
    ap.add_argument("--spawn", action="store_true", help="go through the self-spawn path (one rank per GPU under torch.distributed.run) even for --gpus 1")
    ap.add_argument("--cfg", default="", help="engine / library options, 'name=value,...' (vinet_amd.engine.configure; 'lib.<option>=<int>' reaches vinet_set_option); recorded in the result line")
    ap.add_argument("--force-collectives", action="store_true", help="treat a ONE-rank process group as distributed: every collective of the N > 1 path goes through RCCL (functional test on one GPU)")
    ap.add_argument("--dist-backend", default="", help="process-group backend (default: nccl = RCCL); tests use gloo on device tensors")
    return ap.parse_args(argv)


def host_cores():
    """cores this process may actually use: affinity mask, clipped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(args):
    """The PyTorch-CPU oracle (same aten ops as the reference, fp32) on the host cores:
    a bounded sample (one warm-up + timed steps until ~20 s or --cpu-steps), reported
    beside the GPU number, never the target."""
    from oracle import vinet_cpu as O
    from vinet_amd import synth
    cores = min(host_cores(), 64)   # oneDNN stops scaling (and starts thrashing) far below 256 SMT threads
    torch.set_num_threads(cores)
    av = args.model == "avinet"
    m = (O.VideoAudioSaliencyModel if av else O.VideoSaliencyModel)(num_clips=args.clip)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
    x = synth.clip(1, args.clip, args.height, args.width, 0).permute(0, 2, 1, 3, 4)
    gt = synth.gt_map(1, args.height, args.width, 0)
    inputs = (x, synth.audio(1, 70560, 0)) if av else (x,)
    if args.mode == "train":
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)

        def step():
            opt.zero_grad()
            O.kldiv(m(*inputs), gt).backward()
            opt.step()
    else:
        m.eval()

        def step():
            with torch.no_grad():
                m(*inputs)
    t0 = time.perf_counter()
    step()  # warm-up (oneDNN primitive creation)
    warm = time.perf_counter() - t0
    n, spent = 0, 0.0
    while n < args.cpu_steps and (n == 0 or spent < 20.0) and (warm < 60.0 or n == 0):
        t0 = time.perf_counter()
        step()
        spent += time.perf_counter() - t0
        n += 1
    dt = spent / n
    return dict(value=1.0 / dt, unit="clips/s", cores=cores, kind="port",
                sample="%d %s step(s) of batch 1 at %dx%dx%d fp32 with the PyTorch-CPU oracle (oracle/vinet_cpu.py), "
                       "%d threads, after one warm-up step" % (n, args.mode, args.clip, args.height, args.width, cores))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-run this script as N ranks (one per GPU, RCCL) under
    torch.distributed.run, exactly as the driver's multi-GPU command does; the ranks' stdout (rank 0's JSON line) is ours.
    Refuses (exit code 2) when the node has fewer devices than ranks instead of silently measuring one GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        print("bench.py: --gpus %d but %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        spawn_ranks(args)
    from vinet_amd import _lib, engine, parallel
    parallel.FORCE_COLLECTIVES = bool(args.force_collectives)
    rank, world, local, dev = parallel.init_from_env(args.dist_backend or None)
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let bench.py spawn them)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if parallel.distributed():
        assert dist.get_world_size() == world, "process group size differs from WORLD_SIZE"
        assert dist.get_backend() == "nccl" or args.dist_backend, "RCCL (backend nccl) process group expected"
    _lib.load()
    engine.configure_from_string(args.cfg)
    out = measure(args, rank, world, dev)
    if rank == 0 and world == 1 and args.mode == "train" and not args.no_extras:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out.update(extras(args, dev))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        try:
            out["parity"] = parity_block()
        except Exception as e:      # a report file that cannot be read must never cost the bench line
            out["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    result_line = json.dumps(out) if rank == 0 else None
    # The JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio when its first communicator
    # comes up, and a piped C stream is only flushed at exit -- after Python's own buffer, i.e. behind the result.  Every
    # rank pushes what its C side has buffered out BEFORE the last barrier; rank 0 prints after it.
    def flush_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
    sys.stdout.flush()
    flush_c_stdio()
    if parallel.distributed():
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if result_line is not None:
        print(result_line, flush=True)


def parity_block():
    """what the LAST committed GPU test run measured against the reference's golden vectors (profiles/rN_parity_report.jsonl, written by
    tests/test_gpu_model.py on the GPU box and copied into profiles/): per arithmetic the map error and argmax agreement at the headline
    shape, and the 48-step training-trajectory statistics against the reference's own ensemble.  Read from the file -- nothing is
    re-measured here; `source` names it so the numbers can be checked."""
    import glob
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_parity_report.jsonl")), key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        return None
    last = {}
    with open(files[-1]) as f:
        for line in f:
            try:
                d = json.loads(line)
                last[d["case"]] = d
            except (ValueError, KeyError):
                pass
    out = {"source": "profiles/" + os.path.basename(files[-1]), "contract": "north_star: <= 1e-3 abs on the map, bit-exact argmax", "e2e_32x224x384": {}, "train_trajectory_48_steps": {}}
    for dt in ("fp32", "fp32s", "bf16"):
        e = last.get("e2e_%s_32x224x384" % dt)
        if e:
            out["e2e_32x224x384"][dt] = {"max_abs": e["max_abs"], "argmax_matches": bool(e.get("argmax_matches", dt == "fp32")), "top2_gap": e.get("top2_gap")}
        t = last.get("train_trajectory_" + dt)
        if t:
            out["train_trajectory_48_steps"][dt] = {k: t.get(k) for k in ("early_rel_max", "end_loss", "end_loss_ref_ensemble", "zA", "zE", "eval_after", "eval_after_ref")}
    return out


def _quiet(args, **kw):
    """a copy of the parsed arguments for a secondary leg: no sweep, no site table, no CPU baseline"""
    a = argparse.Namespace(**vars(args))
    a.no_sweep, a.profile_all, a.no_cpu_baseline, a.no_extras = True, False, True, True
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def extras(args, dev):
    """The rest of BASELINE.json's metric (`...; inference fps`, generate_result.py:58-73,98) and the secondary configs, timed
    in the same process after the headline: batch-1 hipGraph fps, batch-64 clips/s, the harness end to end (sliding
    window + post-processing + bytes to the host), AViNet training (config 4, one GPU) and the 64-frame long clip (config 5,
    build-defined decoder tail, no reference parity).  Each leg is a few steps; a failing leg reports its error string."""
    import gc
    import traceback

    def leg(fn):
        try:
            return fn()
        except Exception as e:       # a secondary leg must never take the headline down with it
            traceback.print_exc(file=sys.stderr)
            return {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            gc.collect()
            torch.cuda.empty_cache()

    def infer(batch, graph, steps):
        r = measure(_quiet(args, mode="infer", batch=batch, graph=graph, steps=steps, warmup=2), 0, 1, dev)
        gf, mb = INFER_WORK[(args.clip, args.height, args.width)]
        v = r["value"]
        return {"value": v, "unit": "fps (one output frame per model call)" if batch == 1 else "clips/s", "batch": batch, "hipgraph": bool(graph),
                "ms_per_call": r["ms_per_step"], "hbm_frac": v * mb / 1e3 / HBM_PEAK_GBS, "mfma_frac": v * gf / 1e3 / MFMA_BF16_PEAK_TF}

    def harness():
        import time as _t
        from vinet_amd import engine, model, synth
        from vinet_amd.generate_result import predict_stream, sliding_window_schedule
        engine.set_default_dtype(args.dtype)
        m = model.VideoSaliencyModel(num_clips=args.clip)
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
        m = m.to(dev).eval()
        n = 6 * args.clip - 1
        frames = synth.clip(1, n, args.height, args.width, 0)[0].to(dev)

        def run(fr):     # frames arrive 32 at a time, clips are ring views, maps leave the device as bytes
            got = [mp.clone() for _, mp in predict_stream(m, (fr[c:c + 32] for c in range(0, fr.shape[0], 32)), args.clip, 1, (360, 640), True)]
            return torch.cat(got).cpu()
        run(frames[:2 * args.clip - 1])
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        run(frames)
        torch.cuda.synchronize()
        calls = len(sliding_window_schedule(n, args.clip))
        dt = _t.perf_counter() - t0
        gf, mb = INFER_WORK[(args.clip, args.height, args.width)]
        return {"value": calls / dt, "unit": "fps", "frames": n, "model_calls": calls, "hbm_frac": calls / dt * mb / 1e3 / HBM_PEAK_GBS,
                "mfma_frac": calls / dt * gf / 1e3 / MFMA_BF16_PEAK_TF,
                "what": "vinet_amd.generate_result.predict_stream: sliding window + time-flipped early frames, batch 1, hipGraph replay of "
                        "model call + resize to 360x640 + 11x11 blur + uint8, maps copied to the host as bytes"}

    def train_cfg(**kw):
        r = measure(_quiet(args, mode="train", steps=kw.pop("steps", 5), warmup=1, **kw), 0, 1, dev)
        return {"value": r["value"], "unit": "clips/s", "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                "whole_step": r["whole_step"], "peak_hbm_gb": r["config"]["peak_hbm_gb"]}

    res = {}
    if (args.clip, args.height, args.width) in INFER_WORK and args.model == "vinet":
        res["inference"] = {
            "batch1_hipgraph": leg(lambda: infer(1, True, 200)),
            "batch64": leg(lambda: infer(64, False, 6)),
            "harness_end_to_end": leg(harness),
            "note": "hbm_frac / mfma_frac: BASELINE.md forward work per clip (%.2f GFLOP, %.1f MB) x rate over 8 TB/s / 2.5 PFLOP/s" % INFER_WORK[(args.clip, args.height, args.width)],
        }
    if (args.clip, args.height, args.width) == (32, 224, 384) and args.model == "vinet" and args.dtype == "bf16":
        # the configuration that meets north_star's 1e-3 / bit-exact-argmax contract (tests/test_gpu_model.py::
        # test_e2e_fp32_parity_gate): fp32 activations and weights on v_mfma_f32_16x16x4_f32 (1/16 of the bf16 matrix rate)
        res["fp32_path"] = leg(lambda: dict(train_cfg(dtype="fp32", batch=16),
                                            note="exact-parity configuration (fp32 I/O, fp32 MFMA): max |err| vs the reference 3e-6..5e-6, argmax bit-exact on all "
                                                 "five golden shapes; the bf16 headline holds 1e-2 and the reference's fixation within its top-5 pixels "
                                                 "(profiles/r3_parity_report.jsonl)"))
        # the FAST configuration inside the contract: fp32 tensors, convs on three bf16 MFMAs per product over hi / lo halves of both
        # operands (VINET_F32S; tests/test_gpu_model.py::test_e2e_split_bf16_parity_gate holds 1e-4 and the exact argmax)
        res["parity_path"] = leg(lambda: dict(train_cfg(dtype="fp32s", batch=64),
                                              note="fp32 tensors + split-bf16 matrix arithmetic (3 MFMAs per product, 16 significant bits per operand; forward / data "
                                                   "gradients: split halo-tile kernels (conv_ht.h SPLIT) and conv_dma3, weight gradients: three launches of the bf16 streaming kernels over hi / lo planes): "
                                                   "same gate as fp32_path (<= 1e-3 abs, bit-exact argmax on all five goldens + AViNet)"))
    if (args.clip, args.height, args.width) == (32, 224, 384) and args.model == "vinet":
        res["other_configs"] = {
            "avinet_32x224x384_b192": leg(lambda: train_cfg(model="avinet", batch=0)),
            "vinet_64x256x448_b64": leg(lambda: train_cfg(clip=64, height=256, width=448, batch=0)),
        }
    return res


def measure(args, rank, world, dev):
    """one configuration: warm-up, dominant-kernel discovery, K timed steps -> the result dict (rank 0; other ranks get the
    same dict without meaning)"""
    from vinet_amd import engine, loss, model, optim, parallel, synth
    engine.set_default_dtype(args.dtype)
    engine.WGRAD_SIDE_STREAM = not args.no_side_stream
    torch.cuda.reset_peak_memory_stats(dev)      # `peak_hbm_gb` is this leg's own peak, not the largest leg of the process so far

    if args.batch <= 0:
        args.batch = 64 if (args.clip, args.height, args.width) == (64, 256, 448) else 192
    B = args.batch
    av = args.model == "avinet"
    assert not av or (args.clip, args.height, args.width) == (32, 224, 384), "AViNet's bilinear fusion fixes the clip shape"
    m = (model.VideoAudioSaliencyModel if av else model.VideoSaliencyModel)(num_clips=args.clip)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
    m = m.to(dev)
    # synthetic clip as the loaders hand it over: [B,T,3,H,W], permuted like train.py:205
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.randn((B, args.clip, 3, args.height, args.width), generator=g, device=dev).permute(0, 2, 1, 3, 4)
    gt = synth.gt_map(B, args.height, args.width, rank).to(dev)
    inputs = (x, synth.audio(B, 70560, rank).to(dev)) if av else (x,)

    if args.mode == "train":
        m.train()
        opt = optim.Adam(parallel.trainable_parameters(m), lr=1e-4)
        parallel.broadcast_parameters(opt)
        buckets = parallel.GradientBuckets(opt)      # N > 1: bucketed RCCL all-reduce issued from the tape, overlapped with backward
        buckets.timing = True                        # (two HIP events per 25 MB bucket: issue / completion stamps in the result line)

        def train_step(ins, g_):
            opt.zero_grad()
            buckets.begin_step()
            l = loss.kldiv(m(*ins), g_)
            l.backward()
            buckets.finish()
            opt.step()
            return l

        def step():
            return train_step(inputs, gt)
    else:
        m.eval()
        if args.graph:
            from vinet_amd.graph import GraphedInference
            gm = GraphedInference(m, x.contiguous())
            xin = x.contiguous()

            def step():
                return gm(xin)
        else:
            def step():
                with torch.no_grad():
                    return m(*inputs)

    def sync():
        torch.cuda.synchronize()
        if parallel.distributed():
            dist.barrier()
        torch.cuda.synchronize()

    if args.main_priority != 0:
        hp = torch.cuda.Stream(dev, priority=args.main_priority)
        hp.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(hp)
    # ---- warm-up; the last warm-up step is bracketed site by site to find the dominant kernel site
    graphed = args.mode == "infer" and args.graph
    for i in range(max(args.warmup, 1)):
        if i == max(args.warmup, 1) - 1 and not graphed:
            prof = engine.Profiler()
            engine.set_profiler(prof)
        step()
    if graphed:   # per-site events cannot be recorded inside a replayed graph: profile one eager call
        prof = engine.Profiler()
        engine.set_profiler(prof)
        with torch.no_grad():
            m(x)
    table = prof.summary()
    engine.set_profiler(None)
    # group call sites by the kernel that runs them ("kernel | site"); the dominant KERNEL is
    # the one with the largest total time, as rocprofv3 --stats would rank it
    kernels = {}
    for key, v in table.items():
        kn = key.split(" | ")[0]
        k = kernels.setdefault(kn, dict(ms=0.0, count=0, sites=[]))
        k["ms"] += v["ms"]
        k["count"] += v["count"]
        k["sites"].append(key)
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms"])[0]
    # the kernel with the largest total time among those that CARRY SURVEY 8(d) work (convolutions: forward, data and weight
    # gradients -- algorithmic FLOPs > 0): BatchNorm / copy / activation passes have a budget of zero bytes by that convention, so
    # when one of them leads (`dom`), its streaming efficiency is not a roofline fraction of the path.  Bracketed in the timed
    # region beside `dom`.
    def _flops(k):
        return sum(((table[s_]["work"] or {}).get("flops", 0.0)) * table[s_]["count"] for s_ in kernels[k]["sites"])
    workers = [k for k in kernels if _flops(k) > 0]
    work_k = max(workers, key=lambda k: kernels[k]["ms"]) if workers else None
    # per step, from the bracketed warm-up step: passes that SURVEY 8(d) prices at zero bytes (fused by convention) + pools
    def _ms(*prefixes):
        return sum(v["ms"] for k, v in kernels.items() if k.startswith(prefixes))
    zero_budget = dict(bn_bwd_reduce=_ms("vinet_bn_bwd_reduce"), bn_bwd_apply=_ms("vinet_bn_bwd_apply"),
                       bn_finalize=_ms("vinet_bn_finalize", "vinet_bn_bwd_finalize", "vinet_bn_partials_fold", "vinet_bn_fold"),
                       copy_affine=_ms("vinet_copy_affine"), upsample=_ms("vinet_upsample2x"), act_bwd=_ms("vinet_act_bwd"),
                       import_export=_ms("vinet_import_ncdhw", "vinet_export_ncdhw"))
    zero_budget = {k: round(v, 3) for k, v in zero_budget.items()}
    zero_budget["total"] = round(sum(zero_budget.values()), 3)
    zero_budget["pools_ms_budgeted"] = round(_ms("maxpool_fwd_kernel", "maxpool_bwd_kernel"), 3)
    zero_budget["launches"] = {"bn_bwd_reduce": sum(v["count"] for k, v in kernels.items() if k.startswith("vinet_bn_bwd_reduce")),
                               "bn_bwd_apply": sum(v["count"] for k, v in kernels.items() if k.startswith("vinet_bn_bwd_apply")),
                               "copy_affine": sum(v["count"] for k, v in kernels.items() if k.startswith("vinet_copy_affine"))}
    ZERO_BUDGET_PREFIXES = ("vinet_bn_bwd_reduce", "vinet_bn_bwd_apply", "vinet_bn_finalize", "vinet_bn_bwd_finalize", "vinet_bn_partials_fold",
                            "vinet_bn_fold", "vinet_copy_affine", "vinet_upsample2x", "vinet_act_bwd", "vinet_import_ncdhw", "vinet_export_ncdhw")
    # the ten kernels that own the bracketed warm-up step, with everything a reader needs to recompute their fractions: launches per
    # step, average duration, algorithmic bytes / flops per launch (SURVEY 8(d) convention; 0 bytes for the passes it prices as fused
    # -- their own streaming bytes are listed separately), the roof that bounds them and the fraction reached
    top_kernels = []
    for kn, kv in sorted(kernels.items(), key=lambda kv_: -kv_[1]["ms"])[:10]:
        fl = sum(((table[s_]["work"] or {}).get("flops", 0.0)) * table[s_]["count"] for s_ in kv["sites"]) / max(kv["count"], 1)
        by = sum(((table[s_]["work"] or {}).get("bytes", 0.0)) * table[s_]["count"] for s_ in kv["sites"]) / max(kv["count"], 1)
        avg_s_ = kv["ms"] / max(kv["count"], 1) / 1e3
        zb = kn.startswith(ZERO_BUDGET_PREFIXES)
        mf = fl / max(by, 1.0) > MFMA_BF16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
        row = dict(kernel=kn, launches_per_step=kv["count"], avg_us=round(avg_s_ * 1e6, 1), ms_per_step=round(kv["ms"], 3),
                   algorithmic_flops_per_launch=fl, algorithmic_bytes_per_launch=0.0 if zb else by,
                   bound="mfma" if mf else "hbm",
                   frac=(fl / avg_s_ / 1e12 / MFMA_BF16_PEAK_TF) if mf else (0.0 if zb else by / avg_s_ / 1e9 / HBM_PEAK_GBS))
        if zb:
            row.update(section8d_budget_bytes=0, streaming_bytes_per_launch=by, streaming_frac_of_8TBs=by / avg_s_ / 1e9 / HBM_PEAK_GBS)
        top_kernels.append(row)
    if args.profile_all and rank == 0:
        tot = sum(v["ms"] for v in table.values())
        print("---- kernels (one warm-up step, HIP events) ----", file=sys.stderr)
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"]):
            print("%8.3f ms %5.1f%% x%-4d %s" % (v["ms"], 100 * v["ms"] / tot, v["count"], k), file=sys.stderr)
        print("---- call sites ----", file=sys.stderr)
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            w = v["work"] or {}
            tf = w.get("flops", 0) * v["count"] / max(v["ms"], 1e-9) / 1e9
            gb = w.get("bytes", 0) * v["count"] / max(v["ms"], 1e-9) / 1e6
            print("%8.3f ms %5.1f%% x%-3d %7.1f TF/s %7.1f GB/s  %s" % (v["ms"], 100 * v["ms"] / tot, v["count"], tf, gb, k), file=sys.stderr)
        print("sum of bracketed kernel time %.3f ms" % tot, file=sys.stderr)

    # ---- timed region: only the dominant kernel's launches are bracketed (2 events per launch)
    prof = engine.Profiler(only=kernels[dom]["sites"] + (kernels[work_k]["sites"] if work_k and work_k != dom else []))
    engine.set_profiler(None if graphed else prof)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    t1 = time.perf_counter()
    engine.set_profiler(None)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if parallel.distributed():
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed)
    # N > 1 (or --force-collectives): the gradient exchange of the LAST timed step -- per 25 MB bucket when its all-reduce
    # was issued and when it completed (ms from the start of the step), and how much of the exchange lay inside the backward pass
    comm = buckets.timeline() if (args.mode == "train" and parallel.distributed()) else None
    # ... and the same step at the global batches config 3 will really see (train.py:43 defaults to 8; a DHF1K epoch is 600 clips):
    # 64, 256 and 1536 clips over all ranks, i.e. 8 / 32 / 192 per GPU at N = 8
    gsweep = None
    if args.mode == "train" and parallel.distributed() and not args.no_sweep:
        gsweep = {}
        for G in GLOBAL_BATCH_SWEEP:
            b = G // world
            if G % world or b < 1 or b >= B:
                continue
            xs = x[:b].contiguous()
            ins = (xs, inputs[1][:b].contiguous()) if av else (xs,)
            gs = gt[:b].contiguous()
            train_step(ins, gs)
            sync()
            t0s = time.perf_counter()
            for _ in range(args.sweep_steps):
                train_step(ins, gs)
            sync()
            el = torch.tensor([time.perf_counter() - t0s], dtype=torch.float64, device=dev)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            tl = buckets.timeline()
            gsweep[str(G)] = dict(clips_per_s=G * args.sweep_steps / float(el), local_batch=b, ms_per_step=1e3 * float(el) / args.sweep_steps,
                                  allreduce_hidden_frac=None if tl is None else tl["hidden_frac"])
    if graphed:   # roofline of the dominant kernel from the eager profile call
        prof.records = [r for r in engine.Profiler().records]
        domtab = {k: v for k, v in table.items() if k in kernels[dom]["sites"]}
    else:
        alltab = prof.summary()
        domtab = {k: v for k, v in alltab.items() if k in kernels[dom]["sites"]}
        if not sum(v["count"] for v in domtab.values()):
            # the kernel that led the bracketed warm-up step did not run in the timed steps (a first-step-only
            # kernel can lead under counter collection with --warmup 1): report it from the warm-up step
            domtab = {k: v for k, v in table.items() if k in kernels[dom]["sites"]}
    def _stat(tab):
        return dict(ms=sum(v["ms"] for v in tab.values()), count=sum(v["count"] for v in tab.values()),
                    flops=sum((v["work"] or {}).get("flops", 0.0) * v["count"] for v in tab.values()),
                    bytes=sum((v["work"] or {}).get("bytes", 0.0) * v["count"] for v in tab.values()))
    domstat = _stat(domtab)
    workstat = None
    if work_k is not None:
        wt = {k: v for k, v in (table if graphed else alltab).items() if k in kernels[work_k]["sites"]}
        if not sum(v["count"] for v in wt.values()):
            wt = {k: v for k, v in table.items() if k in kernels[work_k]["sites"]}
        workstat = _stat(wt)

    # ---- local-batch sweep of the same step ({1,2,4,8,16,32}), N = 1 training only
    # (each point eager -- one Python-issued launch per kernel -- and as a replayed hipGraph of the step, vinet_amd.graph)
    sweep, sweep_eager = None, None
    if world == 1 and args.mode == "train" and not args.no_sweep and not parallel.distributed():
        from vinet_amd.graph import GraphedTrainStep
        sweep, sweep_eager = {}, {}
        import gc
        gc.collect()
        torch.cuda.empty_cache()      # the headline's 200 GB of cached blocks: a smaller batch must not have to fight them for memory
        for b in SWEEP_BATCHES:
            if b >= B:
                continue
            xs = x[:b].contiguous()
            ins = (xs, inputs[1][:b].contiguous()) if av else (xs,)
            gs = gt[:b].contiguous()
            for _ in range(2):        # (two warm-up steps: the allocator's block set of this batch size is complete after the second)
                train_step(ins, gs)
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            for _ in range(args.sweep_steps):
                train_step(ins, gs)
            torch.cuda.synchronize()
            sweep_eager[str(b)] = b * args.sweep_steps / (time.perf_counter() - t0s)
            gc.collect()
            torch.cuda.empty_cache()
            gstep = GraphedTrainStep(m, opt, loss.kldiv, ins, gs)
            gstep(ins, gs)
            torch.cuda.synchronize()
            n = max(args.sweep_steps, 4)
            t0s = time.perf_counter()
            for _ in range(n):
                gstep(ins, gs)
            torch.cuda.synchronize()
            sweep[str(b)] = b * n / (time.perf_counter() - t0s)
            del gstep
            gc.collect()
            torch.cuda.empty_cache()

    if rank == 0:
        clips = world * B * args.steps
        value = clips / elapsed
        # per-launch figures are averages over the dominant kernel's launches in the timed region
        avg_s = domstat["ms"] / domstat["count"] / 1e3
        flops, byts = domstat["flops"] / domstat["count"], domstat["bytes"] / domstat["count"]
        ai = flops / max(byts, 1.0)
        ridge = MFMA_BF16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai > ridge:
            roof = dict(bound="mfma", achieved=flops / avg_s / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s")
        else:
            roof = dict(bound="hbm", achieved=byts / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (profiles/pmc_traffic.json; separate --pmc runs, gfx950 FETCH_SIZE correction applied); None if the
        # dominant kernel was not in that profile
        traffic, pmc, pmc_blob = None, None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json"), "rb") as f:
                raw = f.read()
                import hashlib
                pmc_blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()[:12]
                pj = json.loads(raw)
                pmc = pj["kernels"].get(dom.replace(" ", ""))
            if pmc is not None and B == pj.get("batch", 32) and args.mode == "train":
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        roof.update(traffic=traffic, kernel=dom, launches=domstat["count"], avg_us=avg_s * 1e6,
                    algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=byts,
                    also={"TFLOP/s": flops / avg_s / 1e12, "GB/s": byts / avg_s / 1e9})
        if workstat is not None and workstat["count"]:
            # the top kernel that carries SURVEY 8(d) work, with ITS fraction of the roof that bounds it
            w_s = workstat["ms"] / workstat["count"] / 1e3
            wf, wb = workstat["flops"] / workstat["count"], workstat["bytes"] / workstat["count"]
            w_mfma = wf / max(wb, 1.0) > ridge
            roof["work_kernel"] = dict(kernel=work_k, launches=workstat["count"], avg_us=w_s * 1e6, bound="mfma" if w_mfma else "hbm",
                                       achieved=(wf / w_s / 1e12) if w_mfma else (wb / w_s / 1e9), peak=MFMA_BF16_PEAK_TF if w_mfma else HBM_PEAK_GBS,
                                       unit="TFLOP/s" if w_mfma else "GB/s",
                                       frac=(wf / w_s / 1e12 / MFMA_BF16_PEAK_TF) if w_mfma else (wb / w_s / 1e9 / HBM_PEAK_GBS),
                                       algorithmic_flops_per_launch=wf, algorithmic_bytes_per_launch=wb)
            try:
                wp = pj["kernels"].get(work_k.replace(" ", ""))
                if wp is not None and B == pj.get("batch", 32) and args.mode == "train":
                    roof["work_kernel"].update(traffic=wp["traffic_bytes_per_launch"], mfma_busy_frac_pmc=wp.get("mfma_busy_frac"))
            except (NameError, KeyError):
                pass
        roof["zero_budget_pass_ms"] = dict(zero_budget, note="per step, HIP events of the bracketed warm-up step: passes SURVEY 8(d) prices at zero bytes (BatchNorm backward, "
                                                              "materialisation copies, upsample, activation backward, layout import / export); pools listed beside them")
        if args.mode == "train" and engine.WGRAD_SIDE_STREAM:
            roof["note"] = "per-launch time measured inside the step, where the weight-gradient stream shares the CUs and HBM; bench.py --no-side-stream --profile-all gives the kernel alone"
        if traffic is not None:
            roof["traffic_source"] = "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes); (2*FETCH+WRITE)*1024 B, L2-miss traffic incl. Infinity-Cache hits"
            roof["mfma_busy_frac_pmc"] = pmc.get("mfma_busy_frac")
            roof["traffic_source_blob"] = pmc_blob       # git blob id of the counter file this line read (goes stale with the kernels)
            roof["traffic_source_build"] = pj.get("build")
        per_gpu = value / world
        work = TRAIN_WORK.get((args.clip, args.height, args.width)) if args.mode == "train" else None
        whole = None
        if work is not None:
            # the headline is the HBM fraction of the WHOLE path (the net's arithmetic intensity, 215 flop/B,
            # is under the 312 flop/B ridge); the MFMA fraction is quoted beside it
            gb_s = per_gpu * (work[1] + STEP_MB_PER_GPU / B) / 1e3
            whole = dict(bound="hbm", achieved=gb_s, peak=HBM_PEAK_GBS, unit="GB/s", frac=gb_s / HBM_PEAK_GBS,
                         mfma_achieved_tflops=per_gpu * work[0] / 1e3, mfma_frac=per_gpu * work[0] / 1e3 / MFMA_BF16_PEAK_TF,
                         algorithmic_mb_per_clip=work[1], algorithmic_gflop_per_clip=work[0], per_step_mb=STEP_MB_PER_GPU,
                         traffic=None)
            try:   # L2-miss bytes of ONE step summed over every kernel, from the committed PMC passes of this command
                if pj.get("step_traffic_bytes") and B == pj.get("batch", 32):
                    whole["traffic"] = pj["step_traffic_bytes"]
            except NameError:
                pass
        # ---- the line's `roofline` object is the WHOLE STEP: SURVEY 8(d)'s algorithmic bytes of the path over the step
        # time against the HBM roof (the net's arithmetic intensity, 215 flop/B, is under the ridge), the MFMA fraction beside it.  The
        # time-dominant kernel moves to `dominant_kernel` -- with an explicit zero budget when it is a pass 8(d) prices as fused,
        # whose streaming efficiency is NOT a fraction of the path's roofline --, the top FLOP-carrying kernel stays in `work_kernel`.
        dom_zero = dom.startswith(ZERO_BUDGET_PREFIXES)
        dominant = dict(roof)
        if dom_zero:
            dominant.update(section8d_budget_bytes=0, streaming_bytes_per_launch=dominant.pop("algorithmic_bytes_per_launch"),
                            streaming_frac_of_8TBs=dominant.pop("frac"), streaming_gb_s=dominant.pop("achieved"), algorithmic_bytes_per_launch=0.0,
                            note_budget="BatchNorm / copy / activation passes are fused (zero bytes) by SURVEY 8(d)'s convention: this is the efficiency "
                                        "of a pass the roofline says should not exist, not a roofline fraction of the path")
        for k_ in ("work_kernel", "zero_budget_pass_ms"):
            dominant.pop(k_, None)
        # counter traffic of the top kernels from the committed PMC passes of this command, where they match this run
        try:
            if B == pj.get("batch", 32) and args.mode == "train":
                for row in top_kernels:
                    pk = pj["kernels"].get(row["kernel"].replace(" ", ""))
                    if pk is not None:
                        row.update(traffic_bytes_per_launch=pk["traffic_bytes_per_launch"], mfma_busy_frac_pmc=pk.get("mfma_busy_frac"))
        except (NameError, KeyError):
            pass
        if whole is not None:
            step_alg_bytes = (work[1] * B + STEP_MB_PER_GPU) * 1e6
            new_roof = dict(bound="hbm", achieved=whole["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=whole["frac"], traffic=whole["traffic"],
                            scope="whole step (forward + kldiv + backward + fused Adam), per GPU", algorithmic_bytes_per_step=step_alg_bytes,
                            algorithmic_flops_per_step=work[0] * B * 1e9, step_ms=1e3 * elapsed / args.steps,
                            traffic_over_algorithmic=(whole["traffic"] / step_alg_bytes) if whole["traffic"] else None,
                            mfma_achieved_tflops=whole["mfma_achieved_tflops"], mfma_frac=whole["mfma_frac"],
                            algorithmic_mb_per_clip=work[1], algorithmic_gflop_per_clip=work[0], per_step_mb=STEP_MB_PER_GPU,
                            dominant_kernel=dominant, work_kernel=roof.get("work_kernel"), zero_budget_pass_ms=roof.get("zero_budget_pass_ms"),
                            top_kernels=top_kernels,
                            note="frac = SURVEY 8(d) algorithmic bytes of the step / step time / 8 TB/s; per-kernel figures are per launch, timed inside "
                                 "the step (HIP events) where the weight-gradient stream shares CUs and HBM")
            if roof.get("traffic_source"):
                new_roof.update(traffic_source=roof["traffic_source"], traffic_source_blob=roof.get("traffic_source_blob"), traffic_source_build=roof.get("traffic_source_build"))
            roof = new_roof
        else:     # (inference legs, shapes without a SURVEY work figure: the dominant kernel's own roofline, as before)
            roof["top_kernels"] = top_kernels
        out = {
            "metric": ("clips/sec training (%dx%dx%d %s)" % (args.clip, args.height, args.width, args.dtype)) if args.mode == "train" else "inference clips/sec (one output frame per clip)",
            "value": value, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("AViNet (SoundNet + bilinear fusion)" if av else "ViNet DHF1K") + " %s, %dx%dx%d %s, %d clip(s)/GPU/step, kldiv + fused Adam, %s"
                                   % ("training" if args.mode == "train" else "inference", args.clip, args.height,
                                      args.width, args.dtype, B,
                                      "1xMI355X" if world == 1 else "%dxMI355X RCCL all-reduce" % world),
                       "global_batch": world * B, "local_batch": B, "clip": [args.clip, args.height, args.width],
                       "parallelism": "dp%d" % world,
                       "engine": engine.config(changed_only=True),
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                       "reserved_hbm_gb": round(torch.cuda.memory_reserved(dev) / 1e9, 1)},
            "roofline": roof,
            "whole_step": None if whole is None else {"hbm_frac_of_8TBs": whole["frac"], "mfma_frac_of_2.5PF": whole["mfma_frac"]},
        }
        if comm is not None:
            out["allreduce"] = dict(comm, payload_mb=round(opt.flat_g.numel() * 4 / 1e6, 1), bucket_mb=25,
                                    note="RCCL all-reduce (SUM) of the flat fp32 gradient buffer in reverse-order buckets, issued from inside "
                                         "backward; hidden_frac = share of the summed all-reduce time that ended before backward did")
        if gsweep:
            out["global_batch_sweep"] = dict(unit="clips/s", steps_each=args.sweep_steps, global_batch=gsweep)
        if sweep is not None:
            sweep[str(B)] = value
            sweep_eager[str(B)] = value
            out["sweep"] = dict(unit="clips/s", steps_each=args.sweep_steps, primary="local_batch_eager", local_batch=sweep, local_batch_eager=sweep_eager,
                                note="local_batch_eager (PRIMARY: the faster schedule from 4 clips on): one Python-issued launch per kernel; "
                                     "local_batch: the step replayed as one hipGraph (vinet_amd.graph.GraphedTrainStep) for batches below the "
                                     "headline's, which is eager")
        return out
    return {}


if __name__ == "__main__":
    main()
