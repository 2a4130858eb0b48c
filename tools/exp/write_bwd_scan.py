"""Times air_write_bwd for chosen glimpse positions (all items identical): how the graph-order
accumulation chains scale with the out-of-range area.  python tools/exp/write_bwd_scan.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
from air import _hip as H  # noqa: E402

STAMPS = os.environ.get("STAMPS") == "1"
if STAMPS:               # phase stamps of workgroup (0,0): -DAIR_STAMPS build in /tmp (on the GPU box)
    import glob
    import subprocess
    PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")
    out = "/tmp/libair_hip_stamps.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-DAIR_STAMPS", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]
                          + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))) + ["-o", out])
    H._LIB = H.load(out)
    H._LIB.air_debug_stamps.restype = C.c_int
    H._LIB.air_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]

p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
B, N, Cc, w = int(os.environ.get("B", 64)), 3, int(os.environ.get("C", 50)), 28
rng = np.random.RandomState(0)
d_recon = torch.tensor(rng.randn(B, Cc * Cc).astype(np.float32), device="cuda")
if os.environ.get("CONST_G") == "1":
    d_recon.fill_(1.0)
vrec = torch.tensor(rng.uniform(0.05, 0.95, (N, B, w * w)).astype(np.float32), device="cuda")
dgen = torch.zeros(N, B, w * w, device="cuda")
dsx = torch.zeros(N, B, 4, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for s, x, y in ((0.3, 0.0, 0.0), (0.3, 0.9, 0.9), (0.15, 0.95, 0.95), (0.9, 0.0, 0.0)):
    att = torch.zeros(N, B, H.ATT_STRIDE, device="cuda")
    att[..., H.ATT_S], att[..., H.ATT_X], att[..., H.ATT_Y], att[..., H.ATT_Z], att[..., H.ATT_MASK] = s, x, y, 0.7, 1.0
    line = "s=%.2f x=%.2f y=%.2f:" % (s, x, y)
    for lit in (0, 1, 2):
        wb = H.WriteBwd(p(d_recon), p(vrec), p(att), p(dgen), p(dsx), B, N, Cc, w, lit, None, None, None, None)
        for _ in range(5):
            H.check(H.lib().air_write_bwd(C.byref(wb), stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            H.check(H.lib().air_write_bwd(C.byref(wb), stream))
        e1.record()
        torch.cuda.synchronize()
        line += "  literal=%d %.1f us" % (lit, e0.elapsed_time(e1) * 20)
        if STAMPS and lit == 2:
            buf = (C.c_ulonglong * 64)()
            H._LIB.air_debug_stamps(buf, 64)
            v = [int(q) for q in buf]
            idx = [i for i in range(40, 56) if v[i]]   # 41 load | 42 runs | 43 terms | 44 pixel loop + chains | 45 corner atomics | 47 tail
            line += "\n      stamps(us) " + " ".join("%d:%.2f" % (i, (v[i] - v[j]) / 100.0) for j, i in zip(idx, idx[1:]))
            line += "\n      feeder wave 0 (us since stamp 44): " + " ".join("%d:%.2f" % (i, (v[i] - v[44]) / 100.0) for i in range(48, 53) if v[i])
    print(line)
