"""Times air_gemm over (M, N, K) scans to separate the fixed cost of a launch from per-K and
per-tile costs:  python tools/gemm_scan.py  (GPU box)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from air import _hip as H
lib = H.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeop(fn, n=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def gemm(M, N, K, tb, prec, tile=(0, 0), bias=True):
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") if tb else torch.randn(K, N, device="cuda")
    Cc = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    g = H.Gemm(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), M, N, K, K, K if tb else N, N, 0, tb,
               b.data_ptr() if bias else None, None, N, None, N, 0.0, 2 if bias else 0, 0, 0, prec)
    g.tile_m, g.tile_n = tile
    keep = (A, B, Cc, b)
    return lambda: (H.check(lib.air_gemm(C.byref(g), s)), keep)[0]


print("empty-ish kernel floor:", end=" ")
print("%.2f us" % timeop(gemm(16, 16, 4, 0, 0)))
for prec in (0, 1):
    print("precision", prec)
    for (M, N) in ((192, 512), (64, 1024), (192, 256)):
        for tb in (0, 1):
            row = []
            for K in (64, 128, 256, 512, 784, 1568):
                row.append("%6.2f" % timeop(gemm(M, N, K, tb, prec)))
            print("  M=%d N=%d %s  K=64..1568: %s" % (M, N, "NT" if tb else "NN", " ".join(row)))
    for tile in ((1, 1), (1, 2), (2, 2), (2, 4), (4, 2)):
        row = ["%6.2f" % timeop(gemm(192, 512, K, 0, prec, tile)) for K in (256, 784)]
        print("  192x512 tile %s K=256,784: %s" % (tile, " ".join(row)))
