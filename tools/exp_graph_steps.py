"""Does capturing several train steps per hipGraph replay remove per-replay overhead?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
images, targets = synthetic_canvases(64, 50, 2, 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", **HP)
for _ in range(3):
    m.training()
torch.cuda.synchronize()

def timeit(fn, n, per):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * per) * 1e6

def eager():
    s = m._stream(); m._train_phase_a(s); m._train_phase_b(s)
print("eager            %.1f us/step" % timeit(eager, 300, 1))
for k in (1, 2, 4, 8):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(k):
            s = m._stream(); m._train_phase_a(s); m._train_phase_b(s)
    print("graph of %d steps  %.1f us/step" % (k, timeit(g.replay, 300 // k + 20, k)))
