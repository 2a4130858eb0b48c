"""What does a fork/join inside a captured hipGraph cost, and do a wide kernel on a side stream and a chain of
narrow latency-bound kernels on the main stream overlap?  Uses the train step's own launches (dependencies are
irrelevant for timing): main = the backward chain after the VAE data gradients (attend_bwd, dh_heads, 2 x BPTT),
side = the grouped weight-gradient launch / the Adam launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am, _hip as H

hp, B = dict(HP), 64
images, targets = synthetic_canvases(B, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", **hp)
for _ in range(3):
    m.training()
torch.cuda.synchronize()
ops = m.train_step_ops()
names = [o.name for o in ops]
print(names)
i0 = names.index("attend_bwd")
chain = ops[i0:i0 + 4]
wgrad, adam = m._wgrad_plain, ops[-1]

def timeit(build, reps=20, inner=10):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        for _ in range(inner):
            build(side)
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3

def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)

def serial(big):
    def b(side):
        for op in chain: op(st())
        big(st())
    return b

def forked(big, pre=0):
    def b(side):
        cur = torch.cuda.current_stream()
        for op in chain[:pre]: op(st())
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            big(C.c_void_p(side.cuda_stream))
        for op in chain[pre:]: op(st())
        cur.wait_stream(side)
    return b

def only_chain(side):
    for op in chain: op(st())

def empty_fork(side):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    for op in chain: op(st())
    cur.wait_stream(side)

print("chain only            %.2f us" % timeit(only_chain))
print("chain + empty fork    %.2f us" % timeit(empty_fork))
for nm, big in (("wgrad", wgrad), ("adam", adam)):
    print("%s only            %.2f us" % (nm, timeit(lambda s: big(st()))))
    print("chain then %s      %.2f us" % (nm, timeit(serial(big))))
    print("chain || %s        %.2f us" % (nm, timeit(forked(big))))
    print("chain || %s (fork after 1 op)  %.2f us" % (nm, timeit(forked(big, 1))))
