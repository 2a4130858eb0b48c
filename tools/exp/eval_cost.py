"""What one evaluation of the training driver costs on the device (training.py: test_model.forward() on the 1 000 held-out
canvases + AIRModel.numeric_summaries(), every 50 iterations): the replayed forward, its launches one by one, the
summaries launch.  python tools/exp/eval_cost.py [bf16|fp32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
im, tg = synthetic_canvases(64, 50, 2, 1)
tim, ttg = synthetic_canvases(B, 50, 2, 2)
dev = "cuda"
tr = am.AIRModel(torch.tensor(im, device=dev), torch.tensor(tg, device=dev), cnn=False, train=True, annealing_schedules=ANNEAL,
                 gemm_precision=prec, **HP)
te = am.AIRModel(torch.tensor(tim, device=dev), torch.tensor(ttg, device=dev), cnn=False, train=False, reuse=True,
                 annealing_schedules=ANNEAL, gemm_precision=prec, **HP)
for _ in range(3):
    tr.training()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("precision %s, %d test canvases" % (prec, B))
print("eager forward: %.1f us" % timed(te.forward))
s = te._stream()
hi, hop = te._begin_host
ops = [(hop if i == hi else op) for i, op in enumerate(te._fwd)] + [te._finalize]
for op in ops:
    print("   %8.1f us  %-40s %s" % (timed(lambda: op(s), 30), op.name, op.kernel))
te.capture_graph()
print("replayed forward: %.1f us" % timed(te.forward))
out = te.numeric_summaries()
print("summaries launch: %.1f us" % timed(lambda: te.numeric_summaries(out)))
