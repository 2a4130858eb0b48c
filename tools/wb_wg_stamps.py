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
STRESS = "--stress" in sys.argv                      # configs[3]: 128x128 canvas, 5 steps, batch 256
if STRESS:
    sys.argv.remove("--stress")
CARRIED = "--carried" in sys.argv                    # backward="reference_carried" (write_bwd_carried_kernel)
if CARRIED:
    sys.argv.remove("--carried")
hp = dict(HP, canvas_size=128, max_steps=5, max_digits=4) if STRESS else dict(HP)
images, targets = synthetic_canvases(256 if STRESS else 64, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", backward="reference_carried" if CARRIED else "reference", **hp)
for _ in range(5):
    m.training()
torch.cuda.synchronize()
s = m._stream()
which = sys.argv[1] if len(sys.argv) > 1 else "write_bwd"
op = [o for o in m._fwd + m._bwd if which in o.name][0]
names = ["setup", "stage_T", "chains+outputs", "feed(wave 0)", "wait for the others", "final"]
if STRESS:
    # per-CU occupancy of the LDS atomic pipe and of the lane rings at the 128x128 configuration: every workgroup stamps its
    # start [0], the time it spent (summed over its four tap passes) computing terms [1], in the short chains [2] and in the
    # corner phase [3], its hardware id [4], pipe mask + corner terms [5] and its end [6]
    for rep in range(2):
        op(s)
        torch.cuda.synchronize()
        n = 1280
        buf = (C.c_ulonglong * (n * 8))()
        H._LIB.air_debug_stamps_wg(buf, n * 8)
        v = np.array(list(buf), dtype=np.int64).reshape(n, 8)
        live = v[:, 6] > 0
        v = v[live]
        t0 = v[:, 0].min()
        start, end = (v[:, 0] - t0) / 100.0, (v[:, 6] - t0) / 100.0
        terms_t, chains_t, corner_t = v[:, 1] / 100.0, v[:, 2] / 100.0, v[:, 3] / 100.0
        hw = v[:, 4]
        cu = ((hw >> 16) & 0xf) * 1000 + (hw & 0xffff)              # (XCC id, HW_ID low bits: wave / SIMD / CU / SH / SE)
        cu = ((hw >> 16) & 0xf) * 4096 + ((hw >> 8) & 0xf) * 16 + ((hw >> 13) & 0x7) * 256   # XCC, CU_ID (bits 11:8), SE_ID (15:13)
        mask, nterms = v[:, 5] & 0xf, v[:, 5] >> 4
        print("launch %d: %d live workgroups of %d, first start -> last end %.1f us" % (rep, len(v), n, end.max()))
        print("  per workgroup (us): total mean %.1f max %.1f | terms %.1f | chains %.1f | corner phase %.1f (max %.1f)" % (
            (end - start).mean(), (end - start).max(), terms_t.mean(), chains_t.mean(), corner_t.mean(), corner_t.max()))
        print("  corner terms per workgroup: mean %.0f max %.0f; corners on the pipe per workgroup: %s" % (
            nterms.mean(), nterms.max(), np.bincount([bin(int(x)).count("1") for x in mask], minlength=5).tolist()))
        cus = np.unique(cu)
        per = []
        for c in cus:
            sel = cu == c
            per.append((sel.sum(), end[sel].max() - start[sel].min(), corner_t[sel].sum(), terms_t[sel].sum() + chains_t[sel].sum(),
                        nterms[sel].sum()))
        per = np.array(per, dtype=np.float64)
        print("  %d CUs seen: workgroups per CU mean %.2f max %d; CU busy span mean %.1f max %.1f us; sum of corner-phase time per CU mean %.1f "
              "max %.1f us; sum of term + chain time per CU mean %.1f us; corner terms per CU mean %.0f max %.0f -> pipe-only bound %.1f us at 4.06 cycles" % (
                  len(cus), per[:, 0].mean(), per[:, 0].max(), per[:, 1].mean(), per[:, 1].max(), per[:, 2].mean(), per[:, 2].max(),
                  per[:, 3].mean(), per[:, 4].mean(), per[:, 4].max(), per[:, 4].max() * 4.06 / 2100.0))
    sys.exit(0)
if CARRIED and not STRESS:
    # stamps of the blocked kernel: [0] start, [1] set-up done, [2] terms + coordinate gradients + barrier, wave 0: [3] corner
    # chunks + combine, [4] its slots' streams, [6] end; [5] wave 8 after its slots' streams, [7] wave 15 at its end
    for rep in range(3):
        op(s)
        torch.cuda.synchronize()
        n = 192
        buf = (C.c_ulonglong * (n * 8))()
        H._LIB.air_debug_stamps_wg(buf, n * 8)
        v = np.array(list(buf), dtype=np.int64).reshape(n, 8) / 100.0
        v = v[v[:, 1] > 0]
        t0 = v[:, 0].min()
        r = v - v[:, :1]
        print("launch %d: %d live workgroups; first start -> last end %.2f us; start skew %.2f us" % (
            rep, len(v), max(v[:, 6].max(), v[:, 7].max()) - t0, v[:, 0].max() - t0))
        for k, nm in ((1, "set-up done"), (2, "terms + theta + barrier"), (3, "wave 0: corner chunks + combine"), (4, "wave 0: slot streams"),
                      (5, "wave 8: slot streams"), (6, "wave 0: end"), (7, "wave 15: end")):
            print("  %-34s mean %6.2f  max %6.2f" % (nm, r[:, k].mean(), r[:, k].max()))
    sys.exit(0)
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
