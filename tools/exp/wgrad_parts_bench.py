"""Times the grouped weight-gradient launch of the b=64 train step in parts (GPU box): all problems, all but
dWx, dWx stored / norm-only, and every problem alone.  Measured (bf16): all 14.0 us, all but dWx 11.2 us, dWx
stored 7.9 / norm-only 6.7 us, any single K = 192 problem (4 .. 104 workgroups) 7-8.6 us."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch, glob, subprocess
from air import _hip as H
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
if flags:
    PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")
    out = "/tmp/libair_hip_exp.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")] + flags
                          + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))) + ["-o", out])
    H._LIB = H.load(out)
    print("built with", flags)
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
STRESS = os.environ.get("STRESS") == "1"       # configs[3]: 128x128 canvas, 5 steps, batch 256
hp = dict(HP)
if STRESS:
    hp.update(canvas_size=128, max_steps=5, max_digits=4)
HP = hp
images, targets = synthetic_canvases(256 if STRESS else 64, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", **HP)
for _ in range(3):
    m.training()
torch.cuda.synchronize()
lib = H.lib()
arr = m._wgrad_arr
n = len(arr)
s = m._stream()
P = lambda t: C.c_void_p(t.data_ptr())
part = m.store.partials
def t(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps
def sub(idx, null_dw=False):
    ps = []
    for i in idx:
        q = arr[i]
        ps.append(H.Wgrad(q.A, q.dY, None if null_dw else q.dW, q.db, q.M, q.N, q.K, q.lda, q.ldb, q.ldc, q.head_pack, q.Hs, q.Hh, q.Hz,
                          q.A16, q.dY16))
    a = (H.Wgrad * len(ps))(*ps)
    return a, len(ps)
ist = torch.zeros(8, dtype=torch.int32, device="cuda")
for prec in ((1,) if STRESS else (1, 0)):
    for name, (a, k) in (("all", sub(range(n))), ("all but dWx", sub(range(n - 1))), ("dWx stored", sub([n - 1])),
                         ("dWx norm-only", sub([n - 1], True))):
        us = t(lambda: H.check(lib.air_wgrad_grouped(a, k, prec, P(part), P(ist), s), "wgrad"))
        print("prec %d  %-14s %4d workgroups  %.2f us" % (prec, name, lib.air_wgrad_num_workgroups(a, k, prec), us))
    for i in range(n):
        a, k = sub([i])
        q = arr[i]
        us = t(lambda: H.check(lib.air_wgrad_grouped(a, k, prec, P(part), P(ist), s), "wgrad"))
        print("prec %d  problem %2d  M %4d N %4d K %3d lda %4d ldb %4d hp %d  %4d workgroups  %.2f us" % (
            prec, i, q.M, q.N, q.K, q.lda, q.ldb, q.head_pack, lib.air_wgrad_num_blocks(a, k), us))
