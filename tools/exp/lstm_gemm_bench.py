"""In-graph cost of the fused LSTM GEMMs (50 dependent launches per replay): AIR_EPI_LSTM_FWD (wide 64-column tiles vs
quad-unit 16-column tiles), AIR_EPI_LSTM_BWD, against plain GEMMs of the same shapes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import torch
from air import _hip as H
lib = H.lib()
dev = "cuda"
f = lambda *s: torch.randn(*s, device=dev) * 0.3
i16 = lambda *s: torch.zeros(*s, dtype=torch.int16, device=dev)

def twin(t):
    tw = torch.empty(t.shape, dtype=torch.int16, device=dev)
    H.check(lib.air_bf16_twin(t.data_ptr(), tw.data_ptr(), t.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return tw

def timeit(g, n=50, reps=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): H.check(lib.air_gemm(C.byref(g), C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(n): H.check(lib.air_gemm(C.byref(g), s))
    for _ in range(3): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    buf = C.create_string_buffer(128); lib.air_gemm_kernel_name(C.byref(g), buf, 128)
    return e0.elapsed_time(e1) / (n * reps) * 1e3, buf.value.decode()

def G(A, B, Cc, M, N, K, lda, ldb, ldc, **kw):
    g = H.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.precision = M, N, K, lda, ldb, ldc, 1
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g

Bn, R = 64, 256
h, Wh, bias, c_prev, slabs = f(Bn, R), f(R, 4 * R), f(4 * R), f(Bn, R), f(4, Bn, 4 * R)
acts, c1, h1, dummy, h16 = f(Bn, 4 * R), f(Bn, R), f(Bn, R), f(Bn, 4 * R), i16(Bn, R)
hA, WB = twin(h), twin(Wh)
keep = [h, Wh, bias, c_prev, slabs, acts, c1, h1, dummy, h16, hA, WB]
for label, env in (("quad", None), ("wide", "1")):
    if env: os.environ["AIR_LSTM_FWD_WIDE"] = env
    else: os.environ.pop("AIR_LSTM_FWD_WIDE", None)
    g = G(h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, bias=bias, addend=slabs, ldadd=4 * R, addend_slabs=4, epi=H.EPI_LSTM_FWD,
          p0=c_prev, q0=acts, q1=c1, q2=h1, q2_16=h16, A16=hA, B16=WB)
    print("LSTM_FWD %-5s %.2f us  %s" % (label, *timeit(g)))
os.environ.pop("AIR_LSTM_FWD_WIDE", None)
for ns in (1, 2, 8):
    sl = f(ns, Bn, 4 * R)
    keep.append(sl)
    g = G(h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, bias=bias, addend=sl, ldadd=4 * R, addend_slabs=ns, epi=H.EPI_LSTM_FWD,
          p0=c_prev, q0=acts, q1=c1, q2=h1, q2_16=h16, A16=hA, B16=WB)
    print("LSTM_FWD quad, %d slab(s) %.2f us" % (ns, timeit(g)[0]))
g = G(h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, addend=slabs, ldadd=4 * R, addend_slabs=4, epi=H.EPI_LSTM_FWD,
      p0=c_prev, q0=acts, q1=c1, q2=h1, A16=hA, B16=WB)
print("LSTM_FWD quad, no bias, no h16 %.2f us" % timeit(g)[0])
g = G(h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, bias=bias, A16=hA, B16=WB)
print("plain 64x1024x256 nn      %.2f us  %s" % timeit(g))
g = G(h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, bias=bias, addend=slabs, ldadd=4 * R, A16=hA, B16=WB)
print("  + one addend slab       %.2f us  %s" % timeit(g))
dgn, dh_heads, dc_in, ds = f(Bn, 4 * R), f(Bn, R), f(Bn, R), f(Bn, 4 * R)
dh, dg, dcp, dg16 = f(Bn, R), f(Bn, 4 * R), f(Bn, R), i16(Bn, 4 * R)
dA = twin(dgn)
g = G(dgn, Wh, dh, Bn, R, 4 * R, 4 * R, 4 * R, R, transB=1, addend=dh_heads, ldadd=R, epi=H.EPI_LSTM_BWD, p0=acts, p1=c_prev, p2=c1, p3=dc_in,
      q0=dg, q1=dcp, q2=ds, i0=1, q0_16=dg16, A16=dA, B16=WB)
print("LSTM_BWD 64x256x1024 nt   %.2f us  %s" % timeit(g))
g = G(dgn, Wh, dh, Bn, R, 4 * R, 4 * R, 4 * R, R, transB=1, addend=dh_heads, ldadd=R, A16=dA, B16=WB)
print("plain 64x256x1024 nt      %.2f us  %s" % timeit(g))
