"""Phase timing of workgroup (0,0) of the sampler kernels: builds a -DAIR_STAMPS variant of the
library in /tmp ON THE GPU BOX, runs one train step and prints the deltas between stamps.
  python tools/phase_stamps.py [op-name-substring ...]"""
import ctypes as C, glob, os, subprocess, sys
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
for fn in ("air_debug_stamps", "air_debug_stamps_wb", "air_debug_stamps_gemm", "air_debug_stamps_gemm_tw", "air_debug_stamps_wgrad"):
    getattr(H._LIB, fn).restype = C.c_int
    getattr(H._LIB, fn).argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
hp = dict(HP)
STRESS = "--stress" in sys.argv                      # configs[3]: 128x128 canvas, 5 steps, batch 256
if STRESS:
    sys.argv.remove("--stress")
    hp.update(canvas_size=128, max_steps=5, max_digits=4)
images, targets = synthetic_canvases(256 if STRESS else 64, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", **hp)
for _ in range(3):
    m.training()
torch.cuda.synchronize()
s = m._stream()
names = sys.argv[1:] or ["write_bwd", "compose", "attend_fwd", "attend_bwd"]
ops = [m._begin] + m._fwd + m._bwd + [m._wgrad_fused]
for want in names:
    for op in ops:
        if want in op.name and not getattr(op, "_seen_%s" % want, False):
            buf = (C.c_ulonglong * 64)()
            for _ in range(3):
                op(s)
            torch.cuda.synchronize()
            base = {"write_bwd": 40, "attend_fwd": 10, "attend_bwd": 20, "compose": 30, "wgrad": 0}.get(want, 56)
            {56: (H._LIB.air_debug_stamps_gemm_tw if "tw_kernel" in op.kernel else H._LIB.air_debug_stamps_gemm),
             0: H._LIB.air_debug_stamps_wgrad, 40: H._LIB.air_debug_stamps_wb}.get(base, H._LIB.air_debug_stamps)(buf, 64)
            v = [int(x) for x in buf]
            idx = [i for i in range(base - (1 if base == 40 else 0), base + (8 if base == 56 else 22 if base == 40 else 10)) if v[i]]
            idx.sort(key=lambda i: v[i])
            d = ["%d:%.2f" % (i, (v[i] - v[j]) / 100.0) for j, i in zip(idx, idx[1:])]
            print("%-14s total %.2f us | deltas(us) %s" % (op.name, (v[idx[-1]] - v[idx[0]]) / 100.0 if idx else 0, " ".join(d)))
            break
