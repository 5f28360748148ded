#!/usr/bin/env python3
"""Per-kernel wave-cycle budget from a `tools/pmc_sq.sh` pass (rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES of one bench.py step):
wait = waves parked on s_waitcnt / barriers, stall = issue stalls (MFMA dependencies, busy pipes), active = issuing; the three
are disjoint and sum to the wave cycles (MI355X_MICROARCH.md, PMC slots).   python tools/pmc_sq_summarize.py [dir] > profiles/..."""
import collections
import csv
import glob
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_sq/pass1"
rows = list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))
agg = collections.defaultdict(collections.Counter)
calls = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        calls[k] += 1
tot = sum(c["SQ_WAVE_CYCLES"] for c in agg.values())
print("%-64s %5s %6s | %6s %6s %6s | %6s %6s %6s" % ("kernel (2 steps of the 192-clip training step)", "calls", "%wave", "wait", "stall", "active", "valu", "lds", "vmem"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = c["SQ_WAVE_CYCLES"]
    if wc < 0.002 * tot:
        continue
    p = lambda n: 100.0 * c[n] / wc
    print("%-64s %5d %5.1f%% | %5.1f%% %5.1f%% %5.1f%% | %5.1f%% %5.1f%% %5.1f%%" % (
        k[:64], calls[k], 100 * wc / tot, p("SQ_WAIT_ANY"), p("SQ_WAIT_INST_ANY"), p("SQ_ACTIVE_INST_ANY"), p("SQ_ACTIVE_INST_VALU"),
        p("SQ_ACTIVE_INST_LDS"), p("SQ_ACTIVE_INST_VMEM")))
