"""Adam split: part 1 (LSTM + heads variables, full width) on the main stream, part 2 (the VAE variables) as a NARROW
launch on a forked branch under the next step's first kernels (x.Wx, 3 LSTM steps, heads, attend_fwd)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from bench import HP, ANNEAL, synthetic_canvases
from air import air_model as am, _hip as H
from air.air_model import _ptr

hp, B = dict(HP), 64
images, targets = synthetic_canvases(B, hp["canvas_size"], hp["max_digits"], 1)
m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                annealing_schedules=ANNEAL, gemm_precision="bf16", **hp)
for _ in range(3):
    m.training()
torch.cuda.synchronize()
ops = m.train_step_ops()
fwd6 = ops[:6]
adam_full = ops[-1]
st_ = m.store
off = st_.offsets["rec0_w"]
lib = H.lib()
npart = m._wgrad_blocks

def adam_range(lo, hi, blocks):
    def f(s):
        H.check(lib.air_adam_clip_step_blocks(C.c_void_p(st_.params.data_ptr() + 4 * lo), C.c_void_p(st_.grads.data_ptr() + 4 * lo),
                                              C.c_void_p(st_.m.data_ptr() + 4 * lo), C.c_void_p(st_.v.data_ptr() + 4 * lo), hi - lo,
                                              _ptr(st_.partials), npart, _ptr(m.dyn), _ptr(st_.istate), 1.0, 0.9, 0.999, 1e-8,
                                              C.c_void_p(st_.params16.data_ptr() + 2 * lo), _ptr(st_.gnorm), blocks, s))
    return f

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

def serial(side):
    adam_full(st())
    for op in fwd6: op(st())

def split(blocks):
    a1, a2 = adam_range(0, off, 0), adam_range(off, st_.n, blocks)
    def b(side):
        cur = torch.cuda.current_stream()
        a1(st())
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            a2(C.c_void_p(side.cuda_stream))
        for op in fwd6: op(st())
        cur.wait_stream(side)
    return b

if len(sys.argv) > 1 and sys.argv[1] == "trace":
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        for _ in range(3):
            split(64)(side)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    sys.exit(0)
print("n %d, VAE part %d (%.0f %%)" % (st_.n, st_.n - off, 100.0 * (st_.n - off) / st_.n))
print("fwd6 only %.2f us" % timeit(lambda s: [op(st()) for op in fwd6]))
print("adam full only %.2f us" % timeit(lambda s: adam_full(st())))
print("adam part1 only %.2f us" % timeit(lambda s: adam_range(0, off, 0)(st())))
for blocks in (32, 64, 128, 256):
    print("adam part2 alone, %d blocks: %.2f us" % (blocks, timeit(lambda s, b=blocks: adam_range(off, st_.n, b)(st()))))
print("serial: adam ; fwd6   %.2f us" % timeit(serial))
for blocks in (32, 48, 64, 96, 128, 256, 0):
    print("split, part 2 on %3d blocks: %.2f us" % (blocks, timeit(split(blocks))))
