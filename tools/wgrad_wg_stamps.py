"""Per-workgroup timeline of the grouped weight-gradient launch at the bench configuration (debug build with -DAIR_STAMPS
made on the GPU box): when every workgroup starts and ends, on which CU, and how long the launch's tail is.
  python tools/wgrad_wg_stamps.py [--stress]"""
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
H._LIB.air_debug_stamps_wgrad_wg.restype = C.c_int
H._LIB.air_debug_stamps_wgrad_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
STRESS = "--stress" in sys.argv
hp = dict(HP, canvas_size=128, max_steps=5, max_digits=4) if STRESS else dict(HP)
images, targets = synthetic_canvases(256 if STRESS else 64, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", backward="reference_carried", **hp)
for _ in range(5):
    m.training()
torch.cuda.synchronize()
s = m._stream()
op = m._wgrad_fused
for rep in range(3):
    # the launch in its place: behind the step's backward, as in the train step
    m._run_forward(s, finalize=False)
    for o in m._bwd:
        o(s)
    op(s)
    torch.cuda.synchronize()
    n = 2048
    buf = (C.c_ulonglong * (n * 4))()
    H._LIB.air_debug_stamps_wgrad_wg(buf, n * 4)
    v = np.array(list(buf), dtype=np.int64).reshape(n, 4)
    v = v[v[:, 1] > 0]
    t0 = v[:, 0].min()
    start, end = (v[:, 0] - t0) / 100.0, (v[:, 1] - t0) / 100.0
    hw, kind = v[:, 2], v[:, 3]
    cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 8) & 0xf) * 16 + ((hw >> 13) & 0x7) * 256   # XCC, CU_ID (bits 11:8), SE_ID (15:13)
    print("launch %d: %d workgroups on %d CUs; first start -> last end %.2f us; last START at %.2f us; start skew p50 %.2f p90 %.2f us"
          % (rep, len(v), len(set(cu.tolist())), end.max(), start.max(), np.percentile(start, 50), np.percentile(start, 90)))
    dur = end - start
    for k in sorted(set(kind.tolist())):
        sel = kind == k
        print("   K = %3d strip %d: %4d workgroups, duration mean %.2f p90 %.2f max %.2f us; start mean %.2f max %.2f; end mean %.2f max %.2f"
              % (k // 16, k % 16, sel.sum(), dur[sel].mean(), np.percentile(dur[sel], 90), dur[sel].max(), start[sel].mean(), start[sel].max(),
                 end[sel].mean(), end[sel].max()))
    per = np.array([(np.sum(cu == c), end[cu == c].max(), dur[cu == c].sum()) for c in sorted(set(cu.tolist()))])
    print("   per CU: workgroups mean %.2f max %d; last end mean %.2f max %.2f us; sum of workgroup durations mean %.2f max %.2f us"
          % (per[:, 0].mean(), per[:, 0].max(), per[:, 1].mean(), per[:, 1].max(), per[:, 2].mean(), per[:, 2].max()))
    late = start > 0.5 * end.max()
    print("   workgroups that START in the second half of the launch: %d (%s)" % (late.sum(), sorted(set((kind[late] // 16).tolist()))))
