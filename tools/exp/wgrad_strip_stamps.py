"""Phase stamps of one strip workgroup (block 600) of the stand-alone dWx launch: builds air_wgrad.hip with -DAIR_STAMPS on
the GPU box.  python tools/exp/wgrad_strip_stamps.py"""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
out = "/tmp/libwgrad_stamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-DAIR_STAMPS", "-DSTRIP_STAMP_BLOCK=%s" % os.environ.get("BLOCK", "600"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
                       os.path.join(PKG, "csrc", "air_wgrad.hip"), "-o", out])
import torch
from air import _hip as H
lib = C.CDLL(out)
M, N, K = 16384, 1024, 256
dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
A = torch.randn(K, M, device=dev); Y = torch.randn(K, N, device=dev)
A16 = A.to(torch.bfloat16).view(torch.int16); Y16 = Y.to(torch.bfloat16).view(torch.int16)
W = torch.zeros(M, N, device=dev); b = torch.zeros(N, device=dev)
arr = (H.Wgrad * 1)(H.Wgrad(p(A), p(Y), p(W), p(b), M, N, K, M, N, N, 0, 0, 0, 0, p(A16), p(Y16)))
part = torch.zeros(4096, device=dev); ist = torch.zeros(8, dtype=torch.int32, device=dev)
lib.air_wgrad_grouped.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
for _ in range(3):
    assert lib.air_wgrad_grouped(C.cast(arr, C.c_void_p), 1, 1, p(part), p(ist), None) == 0
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
lib.air_debug_stamps_wgrad(buf, 64)
v = list(buf)
names = {0: "enter", 1: "A + dY0 staged"}
G = int(os.environ.get("AIR_WGRAD_STRIP", 4))
for j in range(G):
    for k, nm in enumerate(("barrier", "mfma", "Ct in LDS", "stored", "published", "next staged")):
        names[2 + 6 * j + k] = "t%d %s" % (j, nm)
prev = v[0]
for i in range(0, 2 + 6 * G):
    if v[i]:
        print("%2d %-18s +%6.2f us   (%.2f)" % (i, names.get(i, ""), (v[i] - prev) / 100.0, (v[i] - v[0]) / 100.0)); prev = v[i]
