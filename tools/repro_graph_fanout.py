"""Minimal repro (pure torch, no vinet kernels) for the hipGraph fan-out defect behind round 3's "captured step with deferred
weight gradients computes wrong encoder gradients" (VERDICT r3 item 4).

Pattern, as Ctx.flush_deferred issued it with one join per job:

    main:  A (producer of x)                                   B (consumer of x)
             |\\__ event e1 -> side: S1                        ^
             |\\__ event e2 -> side: S2        (no main-stream node between the event records:
             | ...                              A gets one outgoing edge per join)
             \\__________________________________________________/

Captured, the graph has the edges A -> S1, A -> S2 (redundant: S1 -> S2 is there too), ..., A -> B.  Replayed, B may run
before A has finished once A's fan-out is large enough; eager execution and a graph with ONE join (A -> S1 -> S2 ...) are
correct.  Prints, per number of joins, how many of `reps` replays gave B a stale x.

    python tools/repro_graph_fanout.py            # on an MI355X box
"""
import sys

import torch

dev = torch.device("cuda:0")
N = 1 << 22
x = torch.zeros(N, device=dev)
y = torch.zeros(N, device=dev)
tmp = torch.zeros(16, 1 << 16, device=dev)
side = torch.cuda.Stream()


def body(n_joins, once, busy=20_000_000):
    main = torch.cuda.current_stream()
    x.zero_()
    y.zero_()
    torch.cuda._sleep(busy)          # A is slow: a consumer that does not wait for it reads zeros
    x.fill_(1.0)                     # A: last main-stream node before the joins
    if once:
        side.wait_stream(main)
    for i in range(n_joins):
        if not once:
            side.wait_stream(main)   # event record on main + wait on side, no main-stream node in between
        with torch.cuda.stream(side):
            torch.cuda._sleep(200_000)
            tmp[i].fill_(float(i))
    y.copy_(x)                       # B: must see ones
    main.wait_stream(side)


def run(n_joins, once, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(n_joins, once)          # warm-up (allocations, lazy init)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(n_joins, once)
    bad = 0
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize()
        bad += int(float(y.min()) != 1.0)
    return bad


if __name__ == "__main__":
    print("torch", torch.__version__, "hip", torch.version.hip, torch.cuda.get_device_name(0))
    body(7, False)
    torch.cuda.synchronize()
    print("eager, 7 joins: y.min() =", float(y.min()))
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 12):
        print("joins %2d   one join per side kernel: %2d / 20 replays stale     one join for the batch: %2d / 20" % (n, run(n, False), run(n, True)), flush=True)
