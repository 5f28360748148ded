#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/prof_run.sh (gpurun_out/prof_<tag>/) into the committed evidence:
profiles/<tag>_rocprofv3_kernel_stats.csv, <tag>_pmc_summary.json, pmc_traffic.json (read by bench.py)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1_v4"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
out = os.path.join(ROOT, "profiles")


def load(name):
    f = max(glob.glob(os.path.join(base, "pmc_%s*" % name, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # newest run
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


fe, wr, sq = load("FETCH_SIZE"), load("WRITE_SIZE"), load("SQ_VALU_MFMA")
stats_csv = max(glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats_csv, os.path.join(out, tag + "_rocprofv3_kernel_stats.csv"))
for name in ("bench.json", "bench_under_rocprof.json"):
    if os.path.exists(os.path.join(base, name)):
        shutil.copy(os.path.join(base, name), os.path.join(out, "%s_%s" % (tag, name)))
stats = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"])) for r in csv.DictReader(open(stats_csv))}
rows = []
for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2])[:32]:
    f = fe.get(k, {}).get("FETCH_SIZE", [0, 1])
    w = wr.get(k, {}).get("WRITE_SIZE", [0, 1])
    m = sq.get(k, {})
    mf, ga = m.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 1]), m.get("GRBM_GUI_ACTIVE", [0, 1])
    fetch_kb, write_kb = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
    hbm = (2 * fetch_kb + write_kb) * 1024
    rows.append(dict(kernel=k, pct_of_gpu_time=pct, calls=calls, avg_us=round(avg / 1e3, 1),
                     FETCH_SIZE_KB_raw_per_launch=round(fetch_kb, 1), WRITE_SIZE_KB_per_launch=round(write_kb, 1),
                     traffic_bytes_per_launch=int(hbm), traffic_GBps=round(hbm / (avg * 1e-9) / 1e9) if avg else 0,
                     mfma_busy_frac=round((mf[0] / mf[1]) / ((ga[0] / ga[1]) / 8 * 256 * 4), 3) if ga[0] else None))
note = ("rocprofv3 --pmc, one pass per counter group (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
        "GRBM_GUI_ACTIVE), command: python bench.py --steps 1 --warmup 1 --no-cpu-baseline (batch %d); per-launch averages over all "
        "launches of the kernel.  traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes "
        "for 16 B/lane streams (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included, so this is L2-miss (fabric) "
        "traffic, an upper bound on HBM bytes.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 256 CUs * 4 SIMDs). "
        "avg_us from the separate --kernel-trace --stats run of `python bench.py --steps 4 --warmup 2 --no-cpu-baseline`." % batch)
json.dump(dict(note=note, kernels=rows), open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
short = {}
for r in rows:
    k = re.sub(r"^void ", "", r["kernel"])
    k = re.sub(r"\(.*\)$", "", k).replace(" ", "")
    k = k.replace(",false>", ",plain>").replace(",true>", ",pre>")
    k = k.replace("conv_pp_kernel<2,4,256>", "conv_pp_kernel<256>").replace("conv_pp_kernel<4,2,192>", "conv_pp_kernel<192>")
    k = k.replace("conv_wgrad_pp_kernel<true,", "conv_wgrad_pp_kernel<pre,").replace("conv_wgrad_pp_kernel<false,", "conv_wgrad_pp_kernel<plain,")
    k = k.replace("conv_wgrad_pp_kernel<pre,3>", "conv_wgrad_pp_kernel<pre,192>").replace("conv_wgrad_pp_kernel<pre,4>", "conv_wgrad_pp_kernel<pre,256>")
    k = k.replace("conv_wgrad_pp_kernel<plain,3>", "conv_wgrad_pp_kernel<plain,192>").replace("conv_wgrad_pp_kernel<plain,4>", "conv_wgrad_pp_kernel<plain,256>")
    k = re.sub(r"^channel_reduce8_kernel<\w+,1>$", "vinet_bn_bwd_reduce", k)
    k = re.sub(r"^bn_bwd_apply8_kernel<\w+>$", "vinet_bn_bwd_apply", k)
    # row-streaming weight gradients: the library names them by image width (vinet_conv3d_wgrad_kernel_name)
    k = {"conv_wgrad_rsm_kernel<3,48>": "conv_wgrad_rs_kernel<W48>", "conv_wgrad_rsm_kernel<3,24>": "conv_wgrad_rs_kernel<W24>",
         "conv_wgrad_rs_kernel<3>": "conv_wgrad_rs_kernel<W96>", "conv_wgrad_rs_kernel<6>": "conv_wgrad_rs_kernel<W192>",
         "conv_wgrad_rs_kernel<2>": "conv_wgrad_rs_kernel<W64>", "conv_wgrad_rs_kernel<1>": "conv_wgrad_rs_kernel<W32>"}.get(k, k)
    k = re.sub(r"^bn_bwd_reduce8_bf16_kernel<\d+(,\d+)?>$", "vinet_bn_bwd_reduce", k)
    k = re.sub(r"^bn_bwd_apply8_bf16_kernel<\d+(,\d+)?>$", "vinet_bn_bwd_apply", k)
    m = re.match(r"conv_pw_kernel<(\d+),(\w+)>", k)       # the library names the pointwise kernel by its column-tile width
    if m:
        k = "conv_pw_kernel<%d,%s>" % (int(m[1]) * 16, m[2])
    m = re.match(r"conv_wgrad_dma_kernel<(\d+),(\d+),(\d+),(\d+),(\w+)>", k)
    if m:
        k = "conv_wgrad_dma_kernel<%s,%s,%s,%s>" % (m[1], m[2], m[3], m[5])
    short[k] = dict(traffic_bytes_per_launch=r["traffic_bytes_per_launch"], mfma_busy_frac=r["mfma_busy_frac"], avg_us=r["avg_us"])
# L2-miss bytes of ONE step over every kernel (the PMC passes profile 1 warm-up + 1 timed step: half of the total)
def _tot(agg, name):
    return sum(v[name][0] for v in agg.values() if name in v)


step_traffic = int((2 * _tot(fe, "FETCH_SIZE") + _tot(wr, "WRITE_SIZE")) * 1024 / 2)
import subprocess
try:
    build = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    build = None
json.dump(dict(source="profiles/%s_pmc_summary.json" % tag, build=build, batch=batch, step_traffic_bytes=step_traffic, kernels=short),
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("step traffic %.1f GB" % (step_traffic / 1e9))
for r in rows[:14]:
    print("%-72s %5.1f%% %9.1f us  traffic %8.1f MB  mfma busy %s" % (r["kernel"][:72], r["pct_of_gpu_time"], r["avg_us"], r["traffic_bytes_per_launch"] / 1e6, r["mfma_busy_frac"]))
