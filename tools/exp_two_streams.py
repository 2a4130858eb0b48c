"""Experiment: do two independent half-batch train steps overlap on two streams?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am

def make(B, scope, seed):
    images, targets = synthetic_canvases(B, 50, 2, seed)
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                    scope=scope, annealing_schedules=ANNEAL, seed=seed, gemm_precision="fp32", **HP)
    return m

def timeit(fn, n=200, w=30):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

am.reset_default_graph()
for B in (64, 32, 16):
    m = make(B, "one%d" % B, 1); m.capture_graph()
    print("single stream B=%d: %.1f us/step -> %.0f img/s" % (B, timeit(m.training), B / timeit(m.training) * 1e6))
for nstreams, B in ((2, 32), (4, 16), (2, 64), (4, 64)):
    ms = [make(B, "s%d_%d_%d" % (nstreams, B, i), i) for i in range(nstreams)]
    ss = [torch.cuda.Stream() for _ in ms]
    for m, s in zip(ms, ss):
        with torch.cuda.stream(s):
            m.capture_graph()
    torch.cuda.synchronize()
    def step():
        for m, s in zip(ms, ss):
            with torch.cuda.stream(s):
                m.training()
    t = timeit(step)
    print("%d streams x B=%d: %.1f us per combined step -> %.0f img/s" % (nstreams, B, t, nstreams * B / t * 1e6))
