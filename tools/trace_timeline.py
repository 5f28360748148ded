#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV of bench.py and print, for the last training step: wall time, busy time and gaps of the
main stream, the weight-gradient stream's busy time / first launch / tail, the forward / backward split, and the tiny main-stream
launches (BatchNorm finalize: 7 us alone) that took long -- probes of how long a launch waits for a CU beside which kernel of the
other stream.   python tools/trace_timeline.py gpurun_out/<dir>/**/*_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qs = collections.defaultdict(list)
for r in rows:
    qs[r["Queue_Id"]].append(r)
main = max(qs.values(), key=len)
side = max((l for l in qs.values() if l is not main), key=len, default=[])
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
idx = [i for i, r in enumerate(main) if r["Kernel_Name"].startswith("adam")]
step = main[idx[-2] + 1: idx[-1] + 1]
t0, t1 = S(step[0]), E(step[-1])
busy = sum(E(r) - S(r) for r in step)
print("step %.2f ms: main stream %d launches, busy %.2f ms, gaps %.2f ms" % ((t1 - t0) / 1e6, len(step), busy / 1e6, (t1 - t0 - busy) / 1e6))
bw = next((S(r) for r in step if r["Kernel_Name"].startswith("void loss_bwd")), t0)
print("forward %.2f ms, backward %.2f ms" % ((bw - t0) / 1e6, (t1 - bw) / 1e6))
ss = [r for r in side if t0 <= S(r) <= t1]
if ss:
    print("weight-gradient stream: %d launches, busy %.2f ms, first at %.2f ms, last ends %.2f ms; main stream's last kernel before adam ends %.2f ms"
          % (len(ss), sum(E(r) - S(r) for r in ss) / 1e6, (S(ss[0]) - t0) / 1e6, (E(ss[-1]) - t0) / 1e6, (E(step[-2]) - t0) / 1e6))


def side_at(t):
    for r in ss:
        if S(r) <= t <= E(r):
            return r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    return "-"


probes = [(E(r) - S(r), r) for r in step if r["Kernel_Name"].startswith("bn_bwd_finalize") or r["Kernel_Name"].startswith("bn_finalize")]
slow = sorted((p for p in probes if p[0] > 30000), key=lambda p: -p[0])
print("BatchNorm finalize launches: %d, total %.2f ms, %d above 30 us:" % (len(probes), sum(p[0] for p in probes) / 1e6, len(slow)))
for d, r in slow[:12]:
    print("   %7.1f us at %7.2f ms beside %s" % (d / 1e3, (S(r) - t0) / 1e6, side_at(S(r))))
by = collections.Counter()
for r in ss:
    by[r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]] += E(r) - S(r)
print("weight-gradient stream by kernel:", ", ".join("%s %.1f" % (k, v / 1e6) for k, v in by.most_common(8)))
