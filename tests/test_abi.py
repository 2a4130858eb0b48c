"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads,
exports every symbol include/air_hip.h declares, the ctypes structs match the C
layout, argument errors are reported without touching a GPU, and the product
path refuses to run without the HIP library / on CPU tensors (no fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "air_hip.h")


@pytest.fixture(scope="module")
def H():
    import importlib.util
    spec = importlib.util.spec_from_file_location("air_build", os.path.join(ROOT, "tf-attend-infer-repeat_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    from air import _hip
    _hip.lib()
    return _hip


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(air_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(H):
    declared = _declared_symbols()
    assert len(declared) >= 20
    lib = C.CDLL(H.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(H.EXPORTED_SYMBOLS) == declared


def test_abi_version_and_strerror(H):
    assert H.lib().air_abi_version() == H.ABI_VERSION == 5
    assert b"invalid argument" in H.lib().air_strerror(-1)
    assert H.lib().air_strerror(0) == b"success"


def test_struct_layout_matches_c(H, tmp_path):
    """sizeof/offsetof from a C compile of the header == ctypes."""
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "air_hip.h"\nint main(){'
                    'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(air_gemm_t), offsetof(air_gemm_t, bias),'
                    'offsetof(air_gemm_t, aux), offsetof(air_gemm_t, precision), offsetof(air_gemm_t, p0), offsetof(air_gemm_t, q2), sizeof(air_schedule_t),'
                    'sizeof(air_attend_fwd_t), offsetof(air_attend_fwd_t, B), sizeof(air_attend_bwd_t),'
                    'sizeof(air_write_fwd_t), sizeof(air_write_bwd_t)); printf("%zu %zu %zu %zu %zu\\n", sizeof(air_colsum_t),'
                    'sizeof(air_bottleneck_fwd_t), offsetof(air_bottleneck_fwd_t, ldx), sizeof(air_bottleneck_bwd_t), offsetof(air_bottleneck_bwd_t, H));'
                    'printf("%zu %zu %zu %zu %d\\n", sizeof(air_panel_t), offsetof(air_panel_t, K), offsetof(air_panel_t, exclusive), offsetof(air_gemm_t, B16p),'
                    'AIR_MAX_PANELS);'
                    'printf("%zu %zu %zu %zu\\n", sizeof(air_summaries_t), offsetof(air_summaries_t, out), offsetof(air_summaries_t, B),'
                    'sizeof(air_shuffle_batch_t));'
                    'return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [int(x) for x in out]
    exp = [C.sizeof(H.Gemm), H.Gemm.bias.offset, H.Gemm.aux.offset, H.Gemm.precision.offset, H.Gemm.p0.offset, H.Gemm.q2.offset,
           C.sizeof(H.Schedule), C.sizeof(H.AttendFwd), H.AttendFwd.B.offset, C.sizeof(H.AttendBwd),
           C.sizeof(H.WriteFwd), C.sizeof(H.WriteBwd), C.sizeof(H.Colsum),
           C.sizeof(H.BottleneckFwd), H.BottleneckFwd.ldx.offset, C.sizeof(H.BottleneckBwd), H.BottleneckBwd.H.offset,
           C.sizeof(H.Panel), H.Panel.K.offset, H.Panel.exclusive.offset, H.Gemm.B16p.offset, H.MAX_PANELS,
           C.sizeof(H.Summaries), H.Summaries.out.offset, H.Summaries.B.offset, C.sizeof(H.ShuffleBatch)]
    assert got == exp, (got, exp)


def test_argument_errors_without_gpu(H):
    lib = H.lib()
    assert lib.air_gemm(C.byref(H.Gemm()), None) == -1
    assert lib.air_gemm(None, None) == -1
    assert lib.air_attend_fwd(C.byref(H.AttendFwd()), None) == -1
    assert lib.air_write_bwd(C.byref(H.WriteBwd()), None) == -1
    assert lib.air_lstm_gates_fwd(None, None, None, None, None, 4, 4, None) == -1
    assert lib.air_grad_sqnorm(None, 10, None, None, None) == -1
    assert lib.air_colsum(None, 0, None) == -1
    assert lib.air_vae_bottleneck_fwd(C.byref(H.BottleneckFwd()), None) == -1
    assert lib.air_vae_bottleneck_bwd(None, None) == -1
    buf = (C.c_float * 64)()
    base = C.addressof(buf)
    base += (-base) % 16
    A = C.c_void_p(base)
    # a norm-only weight-gradient problem (dW NULL) needs the partial-sum output
    nul = (H.Wgrad * 1)(H.Wgrad(A, A, None, None, 2, 4, 2, 2, 4, 4, 0, 0, 0, 0))
    assert lib.air_wgrad_grouped(nul, 1, 1, None, None, None) == -1
    assert lib.air_wgrad_num_blocks(nul, 1) == 1
    assert lib.air_wgrad_num_workgroups(nul, 1, 1) == 1 and lib.air_wgrad_num_workgroups(nul, 1, 0) == 1
    assert lib.air_wgrad_num_workgroups(nul, 1, 2) == -1 and lib.air_wgrad_num_workgroups(None, 1, 1) == -1
    # strips are a property of the problem table: >= 512 tiles with twins at precision 1 (2 column tiles per workgroup,
    # 4 from 2048 tiles), never at precision 0 or without twins
    T = C.c_void_p(64)
    big = (H.Wgrad * 1)(H.Wgrad(A, A, A, None, 2048, 1024, 128, 2048, 1024, 1024, 0, 0, 0, 0, T, T))
    assert lib.air_wgrad_num_blocks(big, 1) == 512 and lib.air_wgrad_num_workgroups(big, 1, 1) == 256
    assert lib.air_wgrad_num_workgroups(big, 1, 0) == 512
    big4 = (H.Wgrad * 1)(H.Wgrad(A, A, A, None, 16384, 1024, 256, 16384, 1024, 1024, 0, 0, 0, 0, T, T))
    assert lib.air_wgrad_num_workgroups(big4, 1, 1) == 1024
    notw = (H.Wgrad * 1)(H.Wgrad(A, A, A, None, 2048, 1024, 128, 2048, 1024, 1024, 0, 0, 0, 0))
    assert lib.air_wgrad_num_workgroups(notw, 1, 1) == 512
    assert lib.air_optim_num_partials(1000) > 0
    # panel-blocked twins: null buffers, bad descriptors
    one = (H.Panel * 1)(H.Panel(0, 0, 4, 8, 0, 0))
    assert lib.air_panel_shadow(None, A, one, 1, None) == -1
    assert lib.air_panel_shadow(A, A, one, 0, None) == -1
    assert lib.air_adam_clip_step_panels(A, A, A, A, 32, A, 1, A, A, 1.0, 0.9, 0.999, 1e-8, None, one, 1, None, None, None) == -1
    assert lib.air_adam_clip_step_panels(A, A, A, A, 16, A, 1, A, A, 1.0, 0.9, 0.999, 1e-8, None, one, 1, A, None, None) == -1   # 4 x 8 > 16


def test_no_cpu_fallback(H):
    from air import air_model as am
    am.reset_default_graph()
    with pytest.raises(H.AirHipError):
        am.AIRModel(torch.zeros(4, 2500), torch.zeros(4, dtype=torch.int32), cnn=False)
    with pytest.raises(H.AirHipError):
        H.load("/nonexistent/libair_hip.so")
    from air.transformer import transformer
    with pytest.raises(H.AirHipError):
        transformer(torch.zeros(1, 5, 5, 1), torch.zeros(1, 6), (3, 3))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "tf-attend-infer-repeat_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dp, fn)
                assert "/root/reference" not in txt or fn.endswith(".py") and "import" not in \
                    [ln for ln in txt.splitlines() if "/root/reference" in ln][0]


def test_gemm_dispatch_is_pinned_per_shape(H):
    """Which kernel family an air_gemm descriptor reaches (air_gemm_kernel_name is host-only code, no launch):
      * bf16 twins supplied, whole 16-byte pieces          -> gemm_bf16tw_kernel (air_gemm_bf16.hip)
      * fp32 operands, even K / leading dimensions, row-major A  -> the lean kernels gemm_bf16v2 / gemm_f32v2
      * transposed A, or an odd K / leading dimension / 4-byte-aligned operand -> the general fallback kernels
        gemm_bf16_kernel / gemm_f32_kernel (air_gemm.hip; the only users are ragged test shapes and the transA form
        of the ABI -- no launch of the train step at any BASELINE configuration reaches them)."""
    lib = H.lib()
    buf = C.create_string_buffer(128)

    def name(M, N, K, ta=0, tb=0, prec=1, lda=None, ldb=None, twins=False, a_off=0, epi=0, tile=(0, 0)):
        g = H.Gemm()
        base = 1 << 20
        g.A, g.B, g.C = base + a_off, base * 2, base * 3
        g.M, g.N, g.K = M, N, K
        g.lda = lda if lda is not None else (M if ta else K)
        g.ldb = ldb if ldb is not None else (K if tb else N)
        g.ldc, g.transA, g.transB, g.precision, g.epi = N, ta, tb, prec, epi
        g.tile_m, g.tile_n = tile
        if twins:
            g.A16, g.B16 = base * 4, base * 5
        assert lib.air_gemm_kernel_name(C.byref(g), buf, 128) == 0
        return buf.value.decode()

    # the train step's shapes (Cfg-A and the stress batch): twins -> twin kernels, otherwise the lean ones
    for M, N, K, tb in ((192, 320, 256, 0), (192, 512, 784, 0), (192, 784, 512, 1), (64, 256, 1024, 1), (1280, 512, 784, 0)):
        assert name(M, N, K, tb=tb, twins=True).startswith("gemm_bf16tw_kernel<")
        assert name(M, N, K, tb=tb).startswith("gemm_bf16v2_kernel<")
        assert name(M, N, K, tb=tb, prec=0).startswith("gemm_f32v2_kernel<")
    # even-but-not-multiple-of-4 dimensions (Z = 50): still the lean kernels (8-byte loads)
    assert name(192, 100, 256).startswith("gemm_bf16v2_kernel<") and name(192, 256, 50).startswith("gemm_bf16v2_kernel<")
    assert name(192, 100, 256, twins=True).startswith("gemm_bf16v2_kernel<")          # N % 8 != 0: twins unusable
    assert name(192, 256, 50, twins=True).startswith("gemm_bf16v2_kernel<")           # K % 8 != 0
    # the general fallback kernels: transposed A, odd K, odd leading dimension, 4-byte aligned operand
    for kw in (dict(ta=1), dict(K=333), dict(lda=257), dict(a_off=4)):
        args = dict(M=70, N=150, K=256)
        args.update(kw)
        n1, n0 = name(**args), name(prec=0, **args)
        assert n1.startswith("gemm_bf16_kernel<") and n0.startswith("gemm_f32_kernel<"), (kw, n1, n0)


def test_no_undefined_names_in_the_gpu_only_code():
    """The model's code paths need a GPU and never run in the build container: a static pass over the product, the bench
    and the entry points for names that are bound nowhere (tools/undefined_names.py) -- the NameError class of mistakes."""
    import glob
    files = (glob.glob(os.path.join(ROOT, "tf-attend-infer-repeat_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tf-attend-infer-repeat_amd", "*", "*.py")) +
             [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + glob.glob(os.path.join(ROOT, "tests", "*.py")))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "undefined_names.py")] + files, capture_output=True, text=True)
    assert p.returncode == 0, p.stdout
