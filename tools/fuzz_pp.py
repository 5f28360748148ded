#!/usr/bin/env python3
"""Randomised parity screen of the ping-pong kernels against the CPU model of the ABI (tests/abi_emulator.py):
random channel counts / kernel sizes / strides / paddings / slices, every shape forced through conv_pp (both
widths) and conv_wgrad_pp (both row tiles).   python tools/fuzz_pp.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_kernels as T
from vinet_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = T._lib()
bad = 0
for i in range(n):
    k = (rng.choice([1, 1, 2, 3, 5]), rng.choice([1, 3]), rng.choice([1, 3]))
    s = (rng.choice([1, 1, 2, k[0]]), rng.choice([1, 1, 2]), rng.choice([1, 1, 2]))
    p = (rng.choice([0, k[0] // 2]), k[1] // 2, k[2] // 2)
    dims = (rng.randint(1, 3), rng.randint(max(k[0], 2), 9), rng.randint(3, 14), rng.randint(3, 16))
    if (dims[1] + 2 * p[0] - k[0]) < 0:
        continue
    Cin = 8 * rng.randint(1, 40)
    N = 8 * rng.randint(1, 44)
    ex = {}
    if rng.random() < 0.3:
        ex["stats"] = True
    if rng.random() < 0.3:
        ex["act"] = 1
    if rng.random() < 0.2:
        ex["accumulate"] = True
    if rng.random() < 0.2:
        ex["out_ld"], ex["out_coff"] = N + 16, 8
    if rng.random() < 0.2:
        ex["in_ttotal"], ex["in_toff"] = dims[1] + 2, 1
    case = ("fz%d" % i, dims, Cin, N, k, s, p, ex)
    for opt, vals in ((b"pp", (3, 4)),):
        for v in vals:
            lib.vinet_set_option(opt, v)
            try:
                T._run_conv_case(case, E.BF16, forced=True)
            except AssertionError as e:
                bad += 1
                print("FAIL conv", v, case, str(e)[:200], flush=True)
            finally:
                lib.vinet_set_option(opt, 1)
    wcase = ("fw%d" % i, dims, Cin, N, k, s, p, rng.random() < 0.5)
    for v in (3, 4):
        lib.vinet_set_option(b"wgrad_pp", v)
        try:
            T._run_wgrad_case(wcase, E.BF16)
        except AssertionError as e:
            bad += 1
            print("FAIL wgrad", v, wcase, str(e)[:200], flush=True)
        finally:
            lib.vinet_set_option(b"wgrad_pp", 1)
print("fuzz: %d cases, %d failures" % (n, bad))
sys.exit(1 if bad else 0)
