"""Times air_adam_clip_step vs air_adam_clip_step_factored in isolation (GPU box), 200 back-to-back launches each:
  python tools/exp/adam_factored_bench.py [-DFLAG ...]      (-D flags rebuild the library into /tmp first)
Measured (MI355X, n = 4.01 M, factored block 2500 x 1024, B = 64): stored 20-21 us, factored 27-29 us; with the
tile workgroups' tile rebuild compiled out 16.5 us, with only the tile workgroups running 19 us."""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
import torch
from air import _hip as H
if flags:
    out = "/tmp/libair_hip_exp.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")] + flags
                          + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))) + ["-o", out])
    H._LIB = H.load(out)
lib = H.lib()
D, R4, B = 2500, 1024, 64
n = 4011648
dev = "cuda"
p, g, m, v = (torch.randn(n + 4, device=dev) * 0.01 for _ in range(4))
v.abs_()
X, Dg = torch.rand(B, D, device=dev), torch.randn(B, R4, device=dev)
partials = torch.ones(2048, device=dev); dyn = torch.zeros(32, device=dev); dyn[H.DYN_LEARNING_RATE] = 1e-4; dyn[H.DYN_CLIP_NORM] = 1.0
ist = torch.ones(8, dtype=torch.int32, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
fac = (H.Wgrad * 1)(H.Wgrad(P(X), P(Dg), P(g), None, D, R4, B, D, R4, R4, 0, 0, 0, 0))
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps
for prec in (1, 0):
    a = t(lambda: H.check(lib.air_adam_clip_step(P(p), P(g), P(m), P(v), n, P(partials), 1013, P(dyn), P(ist), 1.0, 0.9, 0.999, 1e-8, None, None, s), "adam"))
    b = t(lambda: H.check(lib.air_adam_clip_step_factored(P(p), P(g), P(m), P(v), n, fac, prec, P(partials), 1013, P(dyn), P(ist), 1.0, 0.9, 0.999, 1e-8, None, s), "adamf"))
    print("prec %d: stored %.2f us   factored %.2f us   %s" % (prec, a, b, " ".join(flags)))
