"""Times every op of the train step in isolation: 200 back-to-back launches of the same op
between two HIP events (duration + launch gap), plus subsets of the grouped wgrad."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am, _hip as H

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
hp, B = dict(HP), 64
if len(sys.argv) > 2 and sys.argv[2] == "stress":            # BASELINE configs[3]
    hp.update(canvas_size=128, max_steps=5, max_digits=4)
    B = 256
images, targets = synthetic_canvases(B, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision=prec, **hp)
for _ in range(3):
    m.training()
torch.cuda.synchronize()
s = m._stream()

def timeop(fn, n=200):
    for _ in range(10): fn(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

ops = m.train_step_ops()
tot = 0
for op in ops:
    t = timeop(op); tot += t
    print("%-34s %7.2f us  %8.1f GB/s  %7.2f TF" % (op.name, t, op.nbytes / t * 1e-3, op.flops / t * 1e-6))
print("sum %.1f us over %d ops" % (tot, len(ops)))
# grouped wgrad subsets
arr = m._wgrad_arr
n = len(arr)
for name, idx in (("dWx only", [n - 1]), ("all but dWx", list(range(0, n - 1))), ("heads only", [n - 2]),
                  ("rec0+out", [2, n - 3])):
    sub = (H.Wgrad * len(idx))(*[arr[i] for i in idx])
    fn = lambda st, sub=sub, k=len(idx): H.check(m.lib.air_wgrad_grouped(sub, k, m._prec, None, None, st))
    print("wgrad subset %-14s %7.2f us" % (name, timeop(fn)))
