"""Per-workgroup phase timeline of the graph-order write backward at the bench configuration (debug build with
-DAIR_STAMPS made on the GPU box): which workgroup is the long pole of the launch, and in which phase.
  python tools/wb_wg_stamps.py"""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
out = "/tmp/libair_hip_stamps.so"
src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
from concurrent.futures import ThreadPoolExecutor
flags = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DAIR_STAMPS",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]
objs = ["/tmp/stamps_%s.o" % os.path.basename(f) for f in src]
with ThreadPoolExecutor(8) as ex:
    list(ex.map(lambda fo: subprocess.check_call(flags + ["-c", fo[0], "-o", fo[1]]), zip(src, objs)))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
import torch
from air import _hip as H
H._LIB = H.load(out)
H._LIB.air_debug_stamps_wg.restype = C.c_int
H._LIB.air_debug_stamps_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
images, targets = synthetic_canvases(64, 50, 2, 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", **HP)
for _ in range(5):
    m.training()
torch.cuda.synchronize()
s = m._stream()
which = sys.argv[1] if len(sys.argv) > 1 else "write_bwd"
op = [o for o in m._fwd + m._bwd if which in o.name][0]
names = ["setup", "stage_T", "chains+outputs", "feed(wave 0)", "wait for the others", "final"]
for rep in range(3):
    op(s)
    torch.cuda.synchronize()
    n = 192
    buf = (C.c_ulonglong * (n * 8))()
    H._LIB.air_debug_stamps_wg(buf, n * 8)
    v = np.array(list(buf), dtype=np.int64).reshape(n, 8)[:, :7] / 100.0
    if which != "write_bwd":
        print("launch %d (%s): start skew %.2f us, first start -> last end %.2f us, workgroup total mean %.2f max %.2f" % (
            rep, which, v[:, 0].max() - v[:, 0].min(), v[:, 6].max() - v[:, 0].min(), (v[:, 6] - v[:, 0]).mean(), (v[:, 6] - v[:, 0]).max()))
        continue
    live = v[:, 1] > 0
    v = v[live]
    t0 = v[:, 0].min()
    d = np.diff(v, axis=1)
    tot = v[:, 6] - v[:, 0]
    print("launch %d: %d live workgroups; first start -> last end %.2f us; start skew %.2f us" % (rep, len(v), v[:, 6].max() - t0, v[:, 0].max() - t0))
    print("  workgroup total: mean %.2f  max %.2f" % (tot.mean(), tot.max()))
    for k, nm in enumerate(names):
        print("  %-22s mean %6.2f  max %6.2f" % (nm, d[:, k].mean(), d[:, k].max()))
    order = np.argsort(-tot)
    print("  slowest workgroups, total (chains+outputs, feed phase): " + "  ".join("%.1f (%.1f, %.1f)" % (tot[i], d[i, 2], d[i, 3] + d[i, 4]) for i in order[:10]))
    print("  longest total without the chain phase: %.2f us" % (tot - d[:, 2]).max())
    w = int(np.argmax(v[:, 6]))
    print("  last workgroup to end: started +%.2f, phases %s" % (v[w, 0] - t0, " ".join("%.2f" % x for x in d[w])))
