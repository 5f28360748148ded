#!/usr/bin/env python3
"""Run pytest while a second stream keeps MFMA waves of ANOTHER kernel resident on every SIMD (tools/ubench/vmem_return_probe.hip,
corun_mfma_kernel: few registers, no memory traffic) -- the condition under which round 5's defect shows:

    python tools/pytest_beside_mfma.py tests/test_gpu_kernels.py -q -x -k exact

The exact-arithmetic kernel tests compare bit for bit with the ABI model: any instruction of the library that loses data beside
foreign MFMA waves fails them here.  (Kernel / operator tests only: the co-runner thread launches continuously, which the tests that
capture hipGraphs or time whole steps do not tolerate.)"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest
import torch

pl = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libvmem_return_probe.so"))
pl.corun_mfma_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
stop = False


def feeder():
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    buf = torch.zeros(1024, device="cuda:0")
    blocks = int(os.environ.get("CORUN_BLOCKS", "416"))
    while not stop:
        for _ in range(4):
            pl.corun_mfma_launch(buf.data_ptr(), 100000, blocks, side.cuda_stream)
        side.synchronize()


th = threading.Thread(target=feeder, daemon=True)
th.start()
time.sleep(1.0)
rc = pytest.main(sys.argv[1:])
stop = True
th.join(timeout=30)
sys.exit(int(rc))
