import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, ctypes as C
import test_gpu_kernels as tg
from vinet_amd import engine as E
lib = tg._lib()
lib.vinet_set_option(b"pw", 2)
def cmp(a, b, tol, what=""):
    a, b = a.float(), b.float()
    bad = ((a - b).abs() > tol * max(1.0, float(b.abs().max()))) | ~torch.isfinite(a)
    print(what, "bad", int(bad.sum()), "of", a.numel())
    if bad.any():
        idx = bad.nonzero().flatten()
        print("  first bad flat idx", idx[:20].tolist(), "last", idx[-5:].tolist())
        print("  got", a.flatten()[idx[:8]].tolist(), "exp", b.flatten()[idx[:8]].tolist())
tg._cmp = cmp
names = sys.argv[1:]
for c in tg.PW_CASES:
    if names and c[0] not in names: continue
    ex = dict(c[7]); ex["tline"] = 6
    print("==", c[0], c[1], c[2], c[3], ex)
    for variant in ({}, ):
        e2 = dict(ex); e2.update(variant)
        try:
            tg._run_conv_case(c[:7] + (e2,), E.BF16, forced=True)
        except AssertionError as e:
            print("  assert:", str(e)[:200])
