"""A/B of the GEMM shapes of the train step: fp32 operands (rounded on their way into LDS) vs bf16 twins.
Each shape: a hipGraph of 50 dependent launches (chained through the stream), replayed 20 times -> us per launch
as it costs inside the captured train step."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from air import _hip as H
lib = H.lib()

def twin(t):
    tw = torch.empty(t.shape, dtype=torch.int16, device="cuda")
    H.check(lib.air_bf16_twin(t.data_ptr(), tw.data_ptr(), t.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return tw

def bench(M, N, K, tb, epi=0, tile=(0, 0), ksplit=0):
    A = torch.randn(M, K, device="cuda"); B = (torch.randn(N, K, device="cuda") if tb else torch.randn(K, N, device="cuda")) * 0.05
    A16, B16 = twin(A), twin(B)
    S = lib.air_gemm_slabs(K, ksplit) if ksplit else 1
    Cc = torch.empty(S, M, N, device="cuda"); C16 = torch.empty(M, N, dtype=torch.int16, device="cuda")
    bias = torch.randn(N, device="cuda")
    res = []
    for mode in (("twin",) if tile == (8, 4) else ("fp32", "twin", "twin+C16")):
        g = H.Gemm()
        g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, (K if tb else N), N
        g.transB, g.precision, g.tile_m, g.tile_n, g.ksplit = tb, 1, tile[0], tile[1], ksplit
        if not ksplit:
            g.bias, g.act = bias.data_ptr(), 2
        if mode != "fp32":
            g.B16 = B16.data_ptr()
            if not ksplit: g.A16 = A16.data_ptr()
        if mode == "twin+C16" and not ksplit:
            g.C16 = C16.data_ptr()
        buf = C.create_string_buffer(128); lib.air_gemm_kernel_name(C.byref(g), buf, 128)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            s = C.c_void_p(st.cuda_stream)
            for _ in range(3): H.check(lib.air_gemm(C.byref(g), s))
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(50): H.check(lib.air_gemm(C.byref(g), s))
        for _ in range(3): gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): gr.replay()
        e1.record(); torch.cuda.synchronize()
        res.append((mode, e0.elapsed_time(e1) / 1000 * 1e3, buf.value.decode()))
    print("%5d x %5d x %5d %s ksplit %d: " % (M, N, K, "nt" if tb else "nn", ksplit) + "  ".join("%s %.2f us" % (m, t) for m, t, _ in res) + "   [" + res[-1][2] + "]")

for shape in [(192, 320, 256, 0), (192, 512, 784, 0), (192, 256, 512, 0), (192, 512, 256, 0), (192, 784, 512, 0),
              (192, 512, 784, 1), (192, 256, 512, 1), (192, 512, 256, 1), (192, 784, 512, 1), (192, 256, 320, 1), (64, 256, 1024, 1)]:
    bench(*shape)
bench(64, 1024, 2500, 0, tile=(2, 2), ksplit=4)
bench(256, 1024, 16384, 0, tile=(4, 2), ksplit=4)
bench(256, 1024, 16384, 0, tile=(4, 4), ksplit=4)
bench(256, 1024, 16384, 0, tile=(4, 4), ksplit=8)
bench(256, 1024, 16384, 0, tile=(8, 4), ksplit=8)
bench(64, 1024, 256, 0, epi=1)
bench(1280, 512, 784, 0); bench(1280, 784, 512, 1); bench(1280, 784, 512, 0)
