"""Stand-alone timing of the input-weight gradient problem of configs[3] (dWx = X^T . sum_t dgates: M = 16384, N = 1024,
K = 256, bf16 twins) through air_wgrad_grouped: python tools/exp/wgrad_strip_bench.py  (AIR_WGRAD_STRIP=<g> selects the
strip width, 0 = one tile per workgroup)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from air import _hip as H
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 1024, 256))]
dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
PA, PB = int(os.environ.get("PADA", 0)), int(os.environ.get("PADB", 0))       # leading-dimension padding (elements)
A = torch.randn(K, M + PA, device=dev); Y = torch.randn(K, N + PB, device=dev)
A16 = A.to(torch.bfloat16).view(torch.int16); Y16 = Y.to(torch.bfloat16).view(torch.int16)
W = torch.zeros(M, N, device=dev); b = torch.zeros(N, device=dev)
NOSTORE = os.environ.get("NOSTORE") == "1"          # norm-only problem: tiles computed and squared, nothing stored
arr = (H.Wgrad * 1)(H.Wgrad(p(A), p(Y), None if NOSTORE else p(W), None if os.environ.get('NOBIAS') == '1' else p(b), M, N, K, M + PA, N + PB, N, 0, 0, 0, 0, p(A16), p(Y16)))
lib = H.lib()
nb = lib.air_wgrad_num_blocks(arr, 1)
part = torch.zeros(nb, device=dev); ist = torch.zeros(8, dtype=torch.int32, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    H.check(lib.air_wgrad_grouped(arr, 1, 1, p(part), p(ist), s))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 50
e0.record()
for _ in range(R):
    lib.air_wgrad_grouped(arr, 1, 1, p(part), p(ist), s)
e1.record(); torch.cuda.synchronize()
print("strip=%s workgroups=%d tiles=%d: %.2f us per launch" % (os.environ.get("AIR_WGRAD_STRIP", "default"),
      lib.air_wgrad_num_workgroups(arr, 1, 1), nb, e0.elapsed_time(e1) * 1000 / R))
