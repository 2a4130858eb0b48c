"""Runs one air_gemm configuration 20 times (for rocprofv3 --pmc):  python3 tools/gemm_one.py M N K tb prec tm tn"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from air import _hip as H
M, N, K, tb, prec, tm, tn = (int(v) for v in sys.argv[1:8])
lib = H.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M, K, device="cuda")
B = torch.randn(N, K, device="cuda") if tb else torch.randn(K, N, device="cuda")
Cc = torch.empty(M, N, device="cuda")
b = torch.randn(N, device="cuda")
g = H.Gemm(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), M, N, K, K, K if tb else N, N, 0, tb,
           b.data_ptr(), None, N, None, N, 0.0, 2, 0, 0, prec)
g.tile_m, g.tile_n = tm, tn
for _ in range(20):
    H.check(lib.air_gemm(C.byref(g), s))
torch.cuda.synchronize()
