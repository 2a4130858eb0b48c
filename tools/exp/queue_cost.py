"""ms per train step of the training driver's hipGraph replay under its input-pipeline variants (bf16, 50 steps per replay):
no input work / row gather only / the pipelined queue (picks made on a forked branch) / dequeue + gather in front of every
step.  python tools/exp/queue_cost.py [steps per replay]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am
from multi_mnist import ShuffleBatchQueue
G = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
im, tg = synthetic_canvases(12000, 50, 2, 1)
images, digits = torch.tensor(im, device=dev), torch.tensor(tg, device=dev)
x, t = torch.zeros(64, 2500, device=dev), torch.zeros(64, dtype=torch.int32, device=dev)
x.copy_(images[:64]); t.copy_(digits[:64])


def run(tag, hooks):
    am.reset_default_graph()
    m = am.AIRModel(x, t, cnn=False, train=True, annealing_schedules=ANNEAL, gemm_precision="bf16", **HP)
    between, after = hooks()
    m.capture_graph(steps=G, between_steps=between, after_steps=after)
    for _ in range(3):
        m.training()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        m.training()
    torch.cuda.synchronize()
    print("%-58s %.4f ms per step" % (tag, (time.perf_counter() - t0) / (n * G) * 1e3))


run("no input work", lambda: (None, None))
q1 = ShuffleBatchQueue(images, digits, 64, x, t, seed=1, min_after_dequeue=2000)
q1.picks.copy_(torch.arange(64, dtype=torch.int32, device=dev))      # the batch of the line above, gathered again every step
run("row gather only (the same 64 records every step)", lambda: ((lambda i: q1._gather(q1.picks)), None))
q0 = ShuffleBatchQueue(images, digits, 64, x, t, seed=1, min_after_dequeue=2000)
run("row gather only, picks all zero (64 copies of ONE image: not a floor)", lambda: ((lambda i: q0._gather(q0.picks)), None))
q2 = ShuffleBatchQueue(images, digits, 64, x, t, seed=1, min_after_dequeue=2000)
run("dequeue + gather in front of every step", lambda: (q2.next_batch, None))
q3 = ShuffleBatchQueue(images, digits, 64, x, t, seed=1, min_after_dequeue=10000 if len(im) > 11000 else 2000)
hooks3 = q3.graph_hooks(G)


def run_piped():
    am.reset_default_graph()
    m = am.AIRModel(x, t, cnn=False, train=True, annealing_schedules=ANNEAL, gemm_precision="bf16", **HP)
    m.capture_graph(steps=G, between_steps=hooks3[0], after_steps=hooks3[1])
    for _ in range(3):
        m.training()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        m.training()
    torch.cuda.synchronize()
    print("%-58s %.4f ms per step" % ("one dequeue_many at the head of the replay + a gather per step", (time.perf_counter() - t0) / (n * G) * 1e3))


run_piped()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
q4 = ShuffleBatchQueue(images, digits, 64, x, t, seed=1, min_after_dequeue=10000 if len(im) > 11000 else 2000)
for _ in range(5):
    q4.next_batch()
e0.record()
for _ in range(100):
    q4.next_batch()
e1.record()
torch.cuda.synchronize()
print("dequeue + gather, stand-alone: %.2f us per batch" % (e0.elapsed_time(e1) * 10))
e0.record()
tab = torch.zeros(G, 64, dtype=torch.int32, device=dev)
import ctypes as C
from air import _hip as H
for _ in range(20):
    H.check(H.lib().air_shuffle_batch_dequeue_many(C.byref(q4._sq), G, tab.data_ptr(), q4._s()))
e1.record()
torch.cuda.synchronize()
print("dequeue_many of %d batches: %.2f us" % (G, e0.elapsed_time(e1) * 50))
