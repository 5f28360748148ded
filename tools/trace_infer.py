"""Timeline of the last hipGraph replay in a rocprofv3 kernel trace of `bench.py --mode infer --batch 1 --graph`:
per kernel start offset, duration, gap to the previous kernel's end on the same queue, and the replay's totals.
usage: python tools/trace_infer.py <kernel_trace.csv> [kernels per replay]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a replay = the kernels between two export kernels (the last launch of a model call)
ends = [i for i, r in enumerate(rows) if "export_ncdhw" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1
rep = rows[lo:hi]
t0 = int(rep[0]["Start_Timestamp"])
busy = 0
last_end = {}
print("replay of %d kernels, %.1f us wall" % (len(rep), (int(rep[-1]["End_Timestamp"]) - t0) / 1e3))
cover, cur_end = 0, t0
for r in rep:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "0")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy += e - s
    if e > cur_end:
        cover += e - max(s, cur_end)
        cur_end = e
    print("%9.1f us  +%6.1f us  gap %6.1f  q%-3s wg %5s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:90]))
print("sum of kernel durations %.1f us, time with any kernel running %.1f us" % (busy / 1e3, cover / 1e3))
