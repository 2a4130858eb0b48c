"""GPU parity tests of the individual C-ABI entry points (libair_hip.so) against
the CPU oracle / fp64 references.  All calls go through the C ABI via ctypes."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import air_oracle as ao  # noqa: E402


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from air import _hip
    _hip.lib()
    return _hip


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bf16_round(x):
    return torch.as_tensor(x).to(torch.bfloat16).to(torch.float64).numpy()


def _run_gemm(H, A, B, M, N, K, ta, tb, prec, bias=None, addend=None, aux=None, act=0, actgrad=0,
              aux_scale=0.0, accumulate=0, c_init=None):
    dev = "cuda"
    At, Bt = torch.tensor(A, device=dev), torch.tensor(B, device=dev)
    Ct = torch.tensor(c_init, device=dev) if c_init is not None else torch.full((M, N), float("nan"), device=dev)
    bt = torch.tensor(bias, device=dev) if bias is not None else None
    adt = torch.tensor(addend, device=dev) if addend is not None else None
    axt = torch.tensor(aux, device=dev) if aux is not None else None
    g = H.Gemm(_p(At), _p(Bt), _p(Ct), M, N, K, A.shape[1], B.shape[1], N, ta, tb,
               _p(bt) if bt is not None else None, _p(adt) if adt is not None else None, N,
               _p(axt) if axt is not None else None, N, aux_scale, act, actgrad, accumulate, prec)
    H.check(H.lib().air_gemm(C.byref(g), _stream()), "air_gemm")
    torch.cuda.synchronize()
    return Ct.cpu().numpy()


def _ref_gemm(A, B, ta, tb, prec, bias=None, addend=None, aux=None, act=0, actgrad=0, aux_scale=0.0,
              accumulate=0, c_init=None):
    A64 = _bf16_round(A) if prec else A.astype(np.float64)
    B64 = _bf16_round(B) if prec else B.astype(np.float64)
    opA = A64.T if ta else A64
    opB = B64.T if tb else B64
    v = opA @ opB
    if bias is not None:
        v = v + bias
    if addend is not None:
        v = v + addend
    if act == 1:
        v = np.maximum(v, 0)
    elif act == 2:
        v = np.log1p(np.exp(-np.abs(v))) + np.maximum(v, 0)
    elif act == 3:
        v = 1 / (1 + np.exp(-(v + aux * aux_scale)))
    if actgrad == 1:
        v = v * (aux > 0)
    elif actgrad == 2:
        v = v * (1 - np.exp(-aux.astype(np.float64)))
    if accumulate:
        v = v + c_init
    return v


GEMM_CASES = [
    # M, N, K, ta, tb  (shapes that occur on the AIR path, Cfg-A)
    (64, 1024, 2500, 0, 0),    # hoisted x.Wx
    (64, 1024, 256, 0, 0),     # h.Wh
    (64, 320, 256, 0, 0),      # heads hidden
    (64, 512, 784, 0, 0),      # vae rec1
    (64, 100, 256, 0, 0),      # mean|log_var (N not /16)
    (64, 256, 50, 0, 0),       # gen1 (K not /4)
    (64, 784, 512, 0, 0),      # gen_mean
    (64, 512, 784, 0, 1),      # dgrad through gen_mean
    (64, 50, 256, 0, 1),       # dgrad to z
    (64, 784, 512, 0, 1),      # dgrad to window
    (64, 256, 1024, 0, 1),     # dgrad through Wh
    (784, 512, 192, 1, 0),     # wgrad rec1
    (2500, 1024, 64, 1, 0),    # wgrad Wx
    (50, 256, 192, 1, 0),      # wgrad gen1
    (7, 33, 5, 0, 0),          # ragged
    (256, 512, 784, 0, 0),     # stress batch
]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K,ta,tb", GEMM_CASES)
def test_gemm_plain(H, M, N, K, ta, tb, prec):
    rng = np.random.RandomState(M + N + K)
    A = rng.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32)
    B = rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32)
    got = _run_gemm(H, A, B, M, N, K, ta, tb, prec)
    ref = _ref_gemm(A, B, ta, tb, prec)
    assert not np.isnan(got).any()
    scale = np.sqrt(K)
    tol = 2e-6 if prec == 0 else 2e-5      # same-rounded operands, fp32 accumulate
    assert np.abs(got - ref).max() / scale < tol, np.abs(got - ref).max()


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_epilogues(H, prec):
    rng = np.random.RandomState(0)
    M, N, K = 64, 100, 256
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    B = (rng.uniform(-1, 1, (K, N)) / 8).astype(np.float32)
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    addend = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    aux = rng.uniform(0.01, 2, (M, N)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    tol = 1e-5 if prec == 0 else 2e-4
    for kw in (dict(bias=bias), dict(bias=bias, act=1), dict(bias=bias, act=2),
               dict(bias=bias, act=3, aux=aux, aux_scale=0.3), dict(addend=addend),
               dict(actgrad=1, aux=aux - 1.0), dict(actgrad=2, aux=aux),
               dict(bias=bias, accumulate=1, c_init=c0)):
        got = _run_gemm(H, A, B, M, N, K, 0, 0, prec, **kw)
        ref = _ref_gemm(A, B, 0, 0, prec, **kw)
        assert np.abs(got - ref).max() < tol, (kw.keys(), np.abs(got - ref).max())


def test_gemm_bad_args(H):
    g = H.Gemm()
    assert H.lib().air_gemm(C.byref(g), None) == -1


def test_transformer_generic_matches_oracle(H):
    from air.transformer import transformer
    rng = np.random.RandomState(0)
    B = 16
    U = rng.uniform(0, 1, (B, 50, 50)).astype(np.float32)
    theta = np.zeros((B, 2, 3), np.float32)
    theta[:, 0, 0] = theta[:, 1, 1] = rng.uniform(0.2, 1.2, B)
    theta[:, 0, 2] = rng.uniform(-1, 1, B)
    theta[:, 1, 2] = rng.uniform(-1, 1, B)
    theta[B // 2:, 0, 1] = rng.uniform(-0.3, 0.3, B - B // 2)      # general affine as well
    theta[B // 2:, 1, 0] = rng.uniform(-0.3, 0.3, B - B // 2)
    for (hi, wi), (ho, wo) in (((50, 50), (28, 28)), ((50, 50), (50, 50))):
        Ui = U[:, :hi, :wi]
        ref = ao.transformer(np.ascontiguousarray(Ui), theta, (ho, wo))
        got = transformer(torch.tensor(np.ascontiguousarray(Ui), device="cuda").unsqueeze(3),
                          torch.tensor(theta, device="cuda"), (ho, wo))[..., 0].cpu().numpy()
        # same op order, no FMA, no libm: the axis-aligned half (the model's own thetas: zero shear entries, so
        # x_s = t00 * x_t + 0 * y_t + t02 has ONE rounding path) is bit for bit; with shear the dot product
        # (t00 * x_t + t01 * y_t) + t02 is summed by numpy's matmul in its own order -- an ulp on some coordinates,
        # which the floor() of a tap that lands on a pixel boundary turns into a different pixel pair
        half = B // 2
        assert np.array_equal(got[:half], ref[:half]), float(np.abs(got[:half] - ref[:half]).max())
        assert np.abs(got[half:] - ref[half:]).max() <= 1e-6, np.abs(got[half:] - ref[half:]).max()
        assert (got[half:] == ref[half:]).mean() > 0.98


def test_write_is_adjoint_of_its_backward(H):
    """<W(v), g> == <v, W^T(g)> for the canvas write and its gather-form adjoint
    (checked through d_gen_pre with the sigmoid factor divided out)."""
    dev = "cuda"
    rng = np.random.RandomState(1)
    B, Cc, w, Z = 8, 50, 28, 50
    vrec = torch.tensor(rng.uniform(0.05, 0.95, (B, w * w)).astype(np.float32), device=dev)
    att = torch.zeros(B, H.ATT_STRIDE, device=dev)
    att[:, H.ATT_S] = torch.tensor(rng.uniform(0.25, 0.8, B).astype(np.float32))
    att[:, H.ATT_X] = torch.tensor(rng.uniform(-0.5, 0.5, B).astype(np.float32))
    att[:, H.ATT_Y] = torch.tensor(rng.uniform(-0.5, 0.5, B).astype(np.float32))
    att[:, H.ATT_Z] = 0.7
    att[:, H.ATT_MASK] = 1.0
    ml = torch.zeros(B, 2 * Z, device=dev)
    dyn = torch.zeros(H.DYN_COUNT, device=dev)
    dyn[H.DYN_VAE_PV] = 1.0
    dyn[H.DYN_GRAD_SCALE] = 1.0 / B
    images = torch.tensor(rng.uniform(0, 1, (B, Cc * Cc)).astype(np.float32), device=dev)
    R = torch.zeros(B, Cc * Cc, device=dev)          # clipped canvas (0 <= z*w*v < 1 here: clip is a no-op)
    rec_loss, L, loss_item = (torch.zeros(B, device=dev) for _ in range(3))
    digits = torch.zeros(B, dtype=torch.int32, device=dev)
    dR = torch.zeros(B, Cc * Cc, device=dev)
    wf = H.WriteFwd(_p(vrec), _p(ml), _p(images), _p(dyn), _p(att), _p(R), _p(rec_loss), _p(dR), _p(L),
                    _p(digits), _p(loss_item), B, 1, Cc, w, Z)
    H.check(H.lib().air_write_fwd(C.byref(wf), _stream()))
    g = torch.tensor(rng.uniform(-1, 1, (B, Cc * Cc)).astype(np.float32), device=dev)
    dgen = torch.zeros(B, w * w, device=dev)
    dsx = torch.zeros(B, 4, device=dev)
    wb = H.WriteBwd(_p(g), _p(vrec), _p(att), _p(dgen), _p(dsx), B, 1, Cc, w)
    H.check(H.lib().air_write_bwd(C.byref(wb), _stream()))
    torch.cuda.synchronize()
    assert bool((digits == 1).all())
    lhs = (R.double() * g.double()).sum(1)                       # <z W v, g>
    dv = dgen.double() / (vrec.double() * (1 - vrec.double()))   # z W^T g
    rhs = (dv * vrec.double()).sum(1)
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-4), (lhs, rhs)
    # d z = <W v, g> = lhs / z
    assert torch.allclose(dsx[:, 3].double(), lhs / 0.7, rtol=1e-4, atol=1e-4)
    # oracle check of the forward (literal op order -> bit-level agreement) and of the fused BCE
    theta = np.zeros((B, 2, 3), np.float32)
    s, x, y = (att[:, i].cpu().numpy() for i in (H.ATT_S, H.ATT_X, H.ATT_Y))
    theta[:, 0, 0] = theta[:, 1, 1] = np.float32(1.0) / s
    theta[:, 0, 2], theta[:, 1, 2] = -x / s, -y / s
    ref = np.float32(0.7) * ao.transformer(vrec.cpu().numpy().reshape(B, w, w), theta, (Cc, Cc)).reshape(B, -1)
    ref = np.clip(ref, 0, 1)
    assert np.abs(R.cpu().numpy() - ref).max() <= 1e-6
    r64, x64 = R.double().cpu().numpy(), images.double().cpu().numpy()
    bce = -np.sum(x64 * np.log(r64 + ao.EPS) + (1 - x64) * np.log(1 - r64 + ao.EPS), axis=1)
    np.testing.assert_allclose(rec_loss.cpu().numpy(), bce, rtol=1e-5)
    np.testing.assert_allclose(loss_item.cpu().numpy(), bce, rtol=1e-5)     # all KLs are 0 here


def test_adam_and_norm_match_oracle(H):
    dev = "cuda"
    rng = np.random.RandomState(2)
    n = 10007 * 4
    p = rng.uniform(-1, 1, n).astype(np.float32)
    g = (rng.standard_normal(n) * 0.05).astype(np.float32)
    m0 = (rng.standard_normal(n) * 0.01).astype(np.float32)
    v0 = (rng.uniform(0, 1e-3, n)).astype(np.float32)
    P, G, M, V = (torch.tensor(a, device=dev) for a in (p, g, m0, v0))
    ist = torch.tensor([4, 0, 0, 0], dtype=torch.int32, device=dev)
    dyn = torch.zeros(H.DYN_COUNT, device=dev)
    dyn[H.DYN_LEARNING_RATE], dyn[H.DYN_CLIP_NORM] = 1e-4, 1.0
    part = torch.zeros(H.lib().air_optim_num_partials(n), device=dev)
    gn = torch.zeros(1, device=dev)
    H.check(H.lib().air_grad_sqnorm(_p(G), n, _p(part), _p(ist), _stream()))
    H.check(H.lib().air_adam_clip_step(_p(P), _p(G), _p(M), _p(V), n, _p(part), part.numel(), _p(dyn), _p(ist),
                                       1.0, 0.9, 0.999, 1e-8, None, _p(gn), _stream()))
    torch.cuda.synchronize()
    assert int(ist[0]) == 5
    grads, gnorm = ao.clip_by_global_norm({"w": g}, 1.0)
    pp, mm, vv = ao.adam_step({"w": p.copy()}, grads, {"w": m0.copy()}, {"w": v0.copy()}, 5, 1e-4)
    assert abs(float(gn) - float(gnorm)) / float(gnorm) < 1e-5
    np.testing.assert_allclose(M.cpu().numpy(), mm["w"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(V.cpu().numpy(), vv["w"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(P.cpu().numpy(), pp["w"], rtol=0, atol=2e-7)


def test_step_begin_schedule_and_noise(H):
    dev = "cuda"
    arr = np.zeros(1, dtype=[("slot", "<i4"), ("flags", "<i4"), ("init", "<f4"), ("iters", "<f4"),
                             ("factor", "<f4"), ("vmin", "<f4"), ("vmax", "<f4")])
    arr[0] = (H.DYN_PRIOR_LOG_ODDS, H.SCHED_HAS_MIN | H.SCHED_LOG, 10000.0, 3000.0, 0.1, 1e-9, 0.0)
    sched = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
    dyn = torch.zeros(H.DYN_COUNT, device=dev)
    nn, nu = 200003, 50001
    normals, unif = torch.zeros(nn, device=dev), torch.zeros(nu, device=dev)
    outs = []
    for step in (0, 3000, 39000):
        ist = torch.tensor([step, 0, 0, 0], dtype=torch.int32, device=dev)
        H.check(H.lib().air_step_begin(_p(sched), 1, _p(dyn), _p(ist), _p(normals), nn, _p(unif), nu,
                                       C.c_uint64(1234), None, None, 0, _stream()))
        torch.cuda.synchronize()
        ref = ao.annealed_value(ao.TRAINING_ANNEALING["z_pres_prior_log_odds"], step)
        assert abs(float(dyn[H.DYN_PRIOR_LOG_ODDS]) - float(ref)) < 2e-3, (step, float(dyn[0]), ref)
        outs.append(normals.clone())
    x, u = normals.double(), unif.double()
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01
    assert abs(float((x ** 4).mean()) - 3.0) < 0.1
    # Box-Muller runs on the hardware log / sin / cos: tails, symmetry and the independence of the (cos, sin) pair
    assert bool(torch.isfinite(x).all()) and float(x.abs().max()) < 6.5
    assert abs(float((x.abs() > 3.0).double().mean()) - 0.0027) < 0.0008 and abs(float((x ** 3).mean())) < 0.05
    pairs = x[: (nn // 2) * 2].reshape(-1, 2)
    assert abs(float((pairs[:, 0] * pairs[:, 1]).mean())) < 0.01
    assert abs(float(((pairs ** 2).sum(1) < 2 * 0.6931).double().mean()) - 0.5) < 0.01      # r^2 = -2 ln u: median 2 ln 2
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 0.01
    assert not torch.equal(outs[0], outs[1])         # stream advances with global_step
    ist = torch.tensor([39000, 0, 0, 0], dtype=torch.int32, device=dev)
    H.check(H.lib().air_step_begin(_p(sched), 1, _p(dyn), _p(ist), _p(normals), nn, _p(unif), nu,
                                   C.c_uint64(1234), None, None, 0, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(normals, outs[2])             # and is reproducible


# ---- GEMM: explicit tiles, split-K slabs, fused epilogues ---------------------------------

def _gemm_struct(H, A, B, Cc, M, N, K, lda, ldb, ldc, prec, **kw):
    g = H.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, lda, ldb, ldc
    g.precision = prec
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("tile", [(1, 1), (1, 2), (1, 4), (2, 2), (2, 4), (4, 1), (4, 2)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0)])
def test_gemm_explicit_tiles(H, tile, ta, tb, prec):
    rng = np.random.RandomState(7)
    M, N, K = 70, 150, 333
    A = rng.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32)
    B = rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32)
    At, Bt = torch.tensor(A, device="cuda"), torch.tensor(B, device="cuda")
    Ct = torch.full((M, N), float("nan"), device="cuda")
    g = _gemm_struct(H, At, Bt, Ct, M, N, K, A.shape[1], B.shape[1], N, prec, transA=ta, transB=tb,
                     tile_m=tile[0], tile_n=tile[1])
    H.check(H.lib().air_gemm(C.byref(g), _stream()))
    torch.cuda.synchronize()
    ref = _ref_gemm(A, B, ta, tb, prec)
    assert np.abs(Ct.cpu().numpy() - ref).max() / np.sqrt(K) < (2e-6 if prec == 0 else 2e-5)


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_split_k_slabs(H, prec):
    rng = np.random.RandomState(8)
    M, N, K = 64, 1024, 2500
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    B = rng.uniform(-1, 1, (K, N)).astype(np.float32)
    At, Bt = torch.tensor(A, device="cuda"), torch.tensor(B, device="cuda")
    for ks, tile in ((8, (2, 2)), (4, (1, 1)), (6, (4, 1)), (3, (0, 0))):
        S = H.lib().air_gemm_slabs(K, ks)
        Ct = torch.full((S, M, N), float("nan"), device="cuda")
        g = _gemm_struct(H, At, Bt, Ct, M, N, K, K, N, N, prec, ksplit=ks, tile_m=tile[0], tile_n=tile[1])
        H.check(H.lib().air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        got = Ct.double().sum(0).cpu().numpy()
        ref = _ref_gemm(A, B, 0, 0, prec)
        assert np.abs(got - ref).max() / np.sqrt(K) < (2e-6 if prec == 0 else 2e-5), (ks, tile)


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_fused_lstm_and_reparam_match_unfused(H, prec):
    dev = "cuda"
    rng = np.random.RandomState(9)
    lib = H.lib()
    Bn, R, Z, HT = 64, 256, 50, 320
    f = lambda *s: torch.tensor(rng.uniform(-1, 1, s).astype(np.float32), device=dev)  # noqa: E731
    # ---- LSTM forward: fused vs gemm + air_lstm_gates_fwd
    h, Wh, bias, c_prev = f(Bn, R), f(R, 4 * R) * 0.1, f(4 * R) * 0.1, f(Bn, R)
    slabs = f(3, Bn, 4 * R) * 0.3
    pre = torch.zeros(Bn, 4 * R, device=dev)
    g = _gemm_struct(H, h, Wh, pre, Bn, 4 * R, R, R, 4 * R, 4 * R, prec, bias=bias, addend=slabs.sum(0).contiguous(),
                     ldadd=4 * R)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    acts0, c0, h0 = torch.zeros(Bn, 4 * R, device=dev), torch.zeros(Bn, R, device=dev), torch.zeros(Bn, R, device=dev)
    H.check(lib.air_lstm_gates_fwd(_p(pre), _p(c_prev), _p(acts0), _p(c0), _p(h0), Bn, R, _stream()))
    acts1, c1, h1 = torch.zeros_like(acts0), torch.zeros_like(c0), torch.zeros_like(h0)
    dummy = torch.zeros(Bn, 4 * R, device=dev)
    g = _gemm_struct(H, h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, prec, bias=bias, addend=slabs, ldadd=4 * R,
                     addend_slabs=3, epi=H.EPI_LSTM_FWD, p0=c_prev, q0=acts1, q1=c1, q2=h1)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    torch.cuda.synchronize()
    for a0, a1 in ((acts0, acts1), (c0, c1), (h0, h1)):
        assert float((a0 - a1).abs().max()) < 2e-5
    # ---- LSTM backward: fused vs gemm + air_lstm_gates_bwd
    d_hid, Whid, dh_rec, dc_in = f(Bn, HT), f(R, HT) * 0.1, f(Bn, R), f(Bn, R)
    dh = torch.zeros(Bn, R, device=dev)
    g = _gemm_struct(H, d_hid, Whid, dh, Bn, R, HT, HT, HT, R, prec, transB=1, addend=dh_rec, ldadd=R)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    dg0, dcp0, ds0 = torch.zeros(Bn, 4 * R, device=dev), torch.zeros(Bn, R, device=dev), f(Bn, 4 * R)
    ds1 = ds0.clone()
    H.check(lib.air_lstm_gates_bwd(_p(dh), _p(dc_in), _p(acts0), _p(c_prev), _p(c0), _p(dg0), _p(dcp0), _p(ds0), 1,
                                   Bn, R, _stream()))
    dg1, dcp1 = torch.zeros_like(dg0), torch.zeros_like(dcp0)
    g = _gemm_struct(H, d_hid, Whid, dh, Bn, R, HT, HT, HT, R, prec, transB=1, addend=dh_rec, ldadd=R,
                     epi=H.EPI_LSTM_BWD, p0=acts0, p1=c_prev, p2=c0, p3=dc_in, q0=dg1, q1=dcp1, q2=ds1, i0=1)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    torch.cuda.synchronize()
    for a0, a1 in ((dg0, dg1), (dcp0, dcp1), (ds0, ds1)):
        assert float((a0 - a1).abs().max()) < 2e-5
    # ---- re-parameterisation forward / backward
    e2, Wml, bml, eps = f(Bn, 256), f(256, 2 * Z) * 0.1, f(2 * Z) * 0.1, f(Bn, Z)
    ml0, zs0 = torch.zeros(Bn, 2 * Z, device=dev), torch.zeros(Bn, Z, device=dev)
    g = _gemm_struct(H, e2, Wml, ml0, Bn, 2 * Z, 256, 256, 2 * Z, 2 * Z, prec, bias=bml)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    H.check(lib.air_reparam_fwd(_p(ml0), _p(eps), _p(zs0), Bn, Z, _stream()))
    ml1, zs1 = torch.zeros_like(ml0), torch.zeros_like(zs0)
    g = _gemm_struct(H, e2, Wml, ml1, Bn, 2 * Z, 256, 256, 2 * Z, 2 * Z, prec, bias=bml, epi=H.EPI_REPARAM_FWD,
                     p0=eps, q0=zs1)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    torch.cuda.synchronize()
    assert float((ml0 - ml1).abs().max()) < 2e-5 and float((zs0 - zs1).abs().max()) < 2e-5
    dg1_, G1 = f(Bn, 256), f(Z, 256) * 0.1
    att = torch.zeros(Bn, H.ATT_STRIDE, device=dev)
    att[:, H.ATT_MASK] = torch.tensor((rng.uniform(0, 1, Bn) > 0.3).astype(np.float32))
    dyn = torch.zeros(H.DYN_COUNT, device=dev)
    dyn[H.DYN_VAE_PV], dyn[H.DYN_VAE_PM], dyn[H.DYN_GRAD_SCALE] = 0.9, 0.1, 1.0 / Bn
    dzs = torch.zeros(Bn, Z, device=dev)
    g = _gemm_struct(H, dg1_, G1, dzs, Bn, Z, 256, 256, 256, Z, prec, transB=1)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    dml0 = torch.zeros(Bn, 2 * Z, device=dev)
    H.check(lib.air_reparam_bwd(_p(dzs), _p(ml0), _p(eps), _p(att), _p(dyn), _p(dml0), Bn, Z, _stream()))
    dml1 = torch.zeros_like(dml0)
    g = _gemm_struct(H, dg1_, G1, dml1, Bn, Z, 256, 256, 256, 2 * Z, prec, transB=1, epi=H.EPI_REPARAM_BWD,
                     p0=ml0, p1=eps, p2=att, p3=dyn)
    H.check(lib.air_gemm(C.byref(g), _stream()))
    torch.cuda.synchronize()
    assert float((dml0 - dml1).abs().max()) < 2e-5


# --------------------------------------------------------------------------- grouped weight gradient
WGRAD_CASES = [
    # (M, N, K) of dW[M,N] = A[K,M]^T . dY[K,N]; shapes of Cfg-A plus ragged ones
    [(2500, 1024, 64), (256, 1024, 192), (256, 320, 192), (784, 512, 192), (512, 256, 192), (256, 100, 192),
     (50, 256, 192), (256, 512, 192), (512, 784, 192)],
    [(70, 33, 5), (64, 64, 64), (130, 100, 200)],          # K not /8, K > 192 (second round), N not /4
    [(100, 36, 1280)],                                     # stress-config depth: N*B = 5*256 rows
]


# (problems of >= 512 tiles -- the first case's 2500 x 1024, the last case -- have their column sums db owned by
# block-rows 0..15 instead of block-row 0)
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("shapes", WGRAD_CASES + [[(256, 320, 64), (4096, 2048, 64)]])
def test_wgrad_grouped(H, shapes, prec):
    rng = np.random.RandomState(len(shapes) + prec)
    dev = "cuda"
    keep, probs, refs = [], [], []
    for (M, N, K) in shapes:
        A = rng.randn(K, M).astype(np.float32)
        dY = rng.randn(K, N).astype(np.float32)
        At, Yt = torch.tensor(A, device=dev), torch.tensor(dY, device=dev)
        Wt = torch.full((M, N), float("nan"), device=dev)
        bt = torch.full((N,), float("nan"), device=dev)
        keep += [At, Yt, Wt, bt]
        probs.append(H.Wgrad(_p(At), _p(Yt), _p(Wt), _p(bt), M, N, K, M, N, N, 0, 0, 0, 0))
        A64 = _bf16_round(A) if prec else A.astype(np.float64)
        Y64 = _bf16_round(dY) if prec else dY.astype(np.float64)
        refs.append((A64.T @ Y64, dY.astype(np.float64).sum(0), Wt, bt, K))       # bias sums stay fp32 in both modes
    arr = (H.Wgrad * len(probs))(*probs)
    nblk = H.lib().air_wgrad_num_blocks(arr, len(probs))
    assert nblk > 0
    part = torch.full((nblk,), float("nan"), device=dev)
    ist = torch.zeros(8, dtype=torch.int32, device=dev)
    H.check(H.lib().air_wgrad_grouped(arr, len(probs), prec, _p(part), _p(ist), _stream()), "air_wgrad_grouped")
    torch.cuda.synchronize()
    sq = 0.0
    for ref_w, ref_b, Wt, bt, K in refs:
        tol = 2e-6 * np.sqrt(K) * 4
        np.testing.assert_allclose(Wt.cpu().numpy(), ref_w, rtol=1e-5, atol=tol * np.abs(ref_w).max())
        np.testing.assert_allclose(bt.cpu().numpy(), ref_b, rtol=1e-5, atol=tol * np.abs(ref_b).max())
        sq += float((Wt.double() ** 2).sum() + (bt.double() ** 2).sum())
    # fused global-norm partials: sum of squares of exactly what was stored; step counted once
    assert abs(float(part.double().sum()) - sq) <= 1e-5 * sq
    assert int(ist[H.IST_GLOBAL_STEP]) == 1


@pytest.mark.parametrize("prec", [0, 1])
def test_wgrad_grouped_head_pack(H, prec):
    """The 7 head output units (air_model.py:294-316, 376): unit o owns only its head's hidden segment."""
    rng = np.random.RandomState(7)
    K, Hs, Hh, Hz = 192, 64, 64, 64
    HT, Hmax = 2 * Hs + 2 * Hh + Hz, max(Hs, Hh, Hz)
    d7 = rng.randn(K, 8).astype(np.float32)
    hid = rng.randn(K, HT).astype(np.float32)
    dt, ht = torch.tensor(d7, device="cuda"), torch.tensor(hid, device="cuda")
    wout = torch.full((7, Hmax), float("nan"), device="cuda")
    bout = torch.full((7,), float("nan"), device="cuda")
    pr = H.Wgrad(_p(dt), _p(ht), _p(wout), _p(bout), 8, HT, K, 8, HT, Hmax, 1, Hs, Hh, Hz)
    arr = (H.Wgrad * 1)(pr)
    part = torch.full((H.lib().air_wgrad_num_blocks(arr, 1),), float("nan"), device="cuda")
    H.check(H.lib().air_wgrad_grouped(arr, 1, prec, _p(part), None, _stream()), "air_wgrad_grouped")
    torch.cuda.synchronize()
    D = _bf16_round(d7) if prec else d7.astype(np.float64)
    Hd = _bf16_round(hid) if prec else hid.astype(np.float64)
    full = D.T @ Hd                                           # [8, HT]
    offs = np.cumsum([0, Hs, Hs, Hh, Hh, Hz])
    head = [0, 1, 2, 2, 3, 3, 4]
    got = wout.cpu().numpy()
    for o in range(7):
        h = head[o]
        np.testing.assert_allclose(got[o, :offs[h + 1] - offs[h]], full[o, offs[h]:offs[h + 1]], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(bout.cpu().numpy(), d7.astype(np.float64).sum(0)[:7], rtol=1e-5, atol=1e-4)
    kept = sum(float((wout[o, :offs[head[o] + 1] - offs[head[o]]].double() ** 2).sum()) for o in range(7))
    kept += float((bout.double() ** 2).sum())
    assert abs(float(part.double().sum()) - kept) <= 1e-5 * kept


# --------------------------------------------------------------------------- un-fused entry points of the ABI
def test_bce_fwd_bwd_matches_oracle_formula(H):
    """air_model.py:580-593: clip, Bernoulli cross-entropy with +1e-9 inside the logs, and its gradient
    (Minimum/Maximum pass their gradient at ties)."""
    rng = np.random.RandomState(3)
    B, D = 5, 2500
    x = (rng.rand(B, D) * (rng.rand(B, D) > 0.8)).astype(np.float32)
    R = rng.uniform(-0.2, 1.3, size=(B, D)).astype(np.float32)
    R[0, :7] = [0.0, 1.0, -0.0, 0.5, 1.0000001, -1e-8, 0.25]            # ties and just-outside values
    dyn = torch.zeros(H.DYN_COUNT, device="cuda")
    dyn[H.DYN_GRAD_SCALE] = 1.0 / B
    xt, Rt = torch.tensor(x, device="cuda"), torch.tensor(R, device="cuda")
    rec = torch.full((B, D), float("nan"), device="cuda")
    loss = torch.full((B,), float("nan"), device="cuda")
    dR = torch.full((B, D), float("nan"), device="cuda")
    H.check(H.lib().air_bce_fwd_bwd(_p(xt), _p(Rt), _p(dyn), _p(rec), _p(loss), _p(dR), B, D, _stream()))
    torch.cuda.synchronize()
    r = np.clip(R.astype(np.float64), 0.0, 1.0)
    np.testing.assert_array_equal(rec.cpu().numpy(), np.clip(R, 0.0, 1.0))
    p1, p0 = r.astype(np.float32) + np.float32(1e-9), (np.float32(1.0) - r.astype(np.float32)) + np.float32(1e-9)
    want = -np.sum(x * np.log(p1.astype(np.float64)) + (1 - x) * np.log(p0.astype(np.float64)), axis=1)
    np.testing.assert_allclose(loss.cpu().numpy(), want, rtol=2e-5)
    g = -(x / p1.astype(np.float64) - (1 - x) / p0.astype(np.float64)) / B
    g[(R > 1.0) | (R < 0.0)] = 0.0                                      # clipped away: no gradient
    np.testing.assert_allclose(dR.cpu().numpy(), g, rtol=2e-4, atol=1e-7)      # fp32 difference of two quotients


def test_colsum_problems(H):
    rng = np.random.RandomState(4)
    shapes = [(192, 512, 512), (64, 1024, 1024), (192, 100, 100), (7, 3, 5)]      # rows, cols, ld
    keep, probs, want = [], [], []
    for i, (rows, cols, ld) in enumerate(shapes):
        src = rng.randn(rows, ld).astype(np.float32)
        init = rng.randn(cols).astype(np.float32)
        st, dt = torch.tensor(src, device="cuda"), torch.tensor(init, device="cuda")
        acc = i % 2
        keep += [st, dt]
        probs.append(H.Colsum(_p(st), _p(dt), rows, cols, ld, acc))
        want.append((dt, src[:, :cols].astype(np.float64).sum(0) + (init if acc else 0.0)))
    arr = (H.Colsum * len(probs))(*probs)
    H.check(H.lib().air_colsum(arr, len(probs), _stream()))
    torch.cuda.synchronize()
    for dt, w in want:
        np.testing.assert_allclose(dt.cpu().numpy(), w, rtol=1e-5, atol=1e-4)


def test_heads_out_wgrad_matches_grouped_launch(H):
    """The stand-alone gradient of the 7 head output units equals the head_pack problem of the
    grouped launch and the plain contraction."""
    rng = np.random.RandomState(5)
    K, Hs, Hh, Hz = 192, 64, 48, 16
    HT, Hmax = 2 * Hs + 2 * Hh + Hz, max(Hs, Hh, Hz)
    d7 = rng.randn(K, 8).astype(np.float32)
    hid = rng.randn(K, HT).astype(np.float32)
    dt, ht = torch.tensor(d7, device="cuda"), torch.tensor(hid, device="cuda")
    dw = torch.zeros(7, Hmax, device="cuda")
    db = torch.zeros(7, device="cuda")
    H.check(H.lib().air_heads_out_wgrad(_p(dt), _p(ht), _p(dw), _p(db), K, Hs, Hh, Hz, Hmax, _stream()))
    torch.cuda.synchronize()
    full = d7.astype(np.float64).T @ hid.astype(np.float64)
    offs = np.cumsum([0, Hs, Hs, Hh, Hh, Hz])
    head = [0, 1, 2, 2, 3, 3, 4]
    got = dw.cpu().numpy()
    for o in range(7):
        h = head[o]
        np.testing.assert_allclose(got[o, :offs[h + 1] - offs[h]], full[o, offs[h]:offs[h + 1]], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), d7.astype(np.float64).sum(0)[:7], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("Hi,Wi,Ho,Wo", [(28, 28, 50, 50), (50, 50, 28, 28), (9, 11, 7, 8), (128, 128, 28, 28)])
def test_transformer_bwd_generic_matches_oracle(H, Hi, Wi, Ho, Wo):
    """air_transformer_bwd (any theta: rotation / shear / out-of-range) vs oracle.transformer_backward,
    which is pinned to the reference's executed graph (tests/test_graph_exec.py): d U bit for bit (the
    UnsortedSegmentSum accumulation order), d theta to 2e-5 (a contraction over the output pixels)."""
    from air.transformer import transformer, transformer_grad
    rng = np.random.RandomState(Hi * 7 + Wo)
    B = 5
    U = rng.uniform(0, 1, (B, Hi, Wi)).astype(np.float32)
    base = np.array([[0.55, 0.25, 0.3], [-0.2, 0.6, -0.4]], np.float32)
    th = (np.tile(base, (B, 1, 1)) + rng.randn(B, 2, 3).astype(np.float32) * 0.15).astype(np.float32)
    th[0] = [[1.6, 0.0, 0.9], [0.0, 1.6, -0.9]]              # mostly out of range: border slots collect long chains
    d = (rng.randn(B, Ho, Wo) * np.where(rng.uniform(size=(B, Ho, Wo)) < 0.1, 1e6, 1.0)).astype(np.float32)
    dU_ref, dth_ref = ao.transformer_backward(U, th, (Ho, Wo), d)
    Ut, tt, dt = (torch.tensor(v, device="cuda") for v in (U, th, d))
    dU, dth = transformer_grad(Ut, tt, (Ho, Wo), dt)
    torch.cuda.synchronize()
    assert np.array_equal(dU.cpu().numpy(), dU_ref)
    scale = np.abs(dth_ref).max(axis=(1, 2), keepdims=True)
    assert (np.abs(dth.cpu().numpy().reshape(B, 2, 3) - dth_ref) <= 2e-5 * scale).all()
    # through torch.autograd (the drop-in op is differentiable like the reference's under tf.gradients)
    Ug, tg = Ut.clone().requires_grad_(True), tt.clone().requires_grad_(True)
    transformer(Ug, tg, (Ho, Wo)).backward(dt)
    assert torch.equal(Ug.grad, dU) and torch.equal(tg.grad.reshape(B, 6), dth)


# ---- exported un-fused kernels vs the oracle directly (not only HIP-vs-HIP) ------------------------------

def test_unfused_lstm_and_reparam_kernels_match_oracle(H):
    """air_lstm_gates_fwd/bwd, air_lstm_first_step, air_reparam_fwd/bwd against oracle.lstm_cell / oracle.vae
    arithmetic (BasicLSTMCell: i, j, f, o, forget bias 1.0 -- air_model.py:286; vae.py:22-24) and their
    torch-autograd gradients."""
    from oracle import air_oracle_torch as at  # noqa: F401
    dev, lib = "cuda", H.lib()
    rng = np.random.RandomState(21)
    Bn, D, R, Z = 37, 19, 24, 50
    x, h, c = (rng.randn(Bn, k).astype(np.float32) for k in (D, R, R))
    K = (rng.randn(D + R, 4 * R) * 0.3).astype(np.float32)
    b = (rng.randn(4 * R) * 0.1).astype(np.float32)
    c_ref, h_ref = ao.lstm_cell(x, c, h, K, b)
    pre = (np.concatenate([x, h], 1) @ K + b).astype(np.float32)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    pre_d, c_d = t(pre), t(c)
    acts, c1, h1 = torch.zeros(Bn, 4 * R, device=dev), torch.zeros(Bn, R, device=dev), torch.zeros(Bn, R, device=dev)
    H.check(lib.air_lstm_gates_fwd(_p(pre_d), _p(c_d), _p(acts), _p(c1), _p(h1), Bn, R, _stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(c1.cpu().numpy(), c_ref, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), h_ref, rtol=2e-6, atol=2e-6)
    # first step: zero state, pre-activation = sum of the x.Wx slabs (in slab order) + bias
    slabs = (rng.randn(5, Bn, 4 * R) * 0.4).astype(np.float32)
    s = np.zeros((Bn, 4 * R), np.float32)
    for k in range(5):
        s = s + slabs[k]
    c0_ref, h0_ref = ao.lstm_cell(np.zeros((Bn, 1), np.float32), np.zeros((Bn, R), np.float32), np.zeros((Bn, R), np.float32),
                                  np.zeros((1 + R, 4 * R), np.float32), np.zeros(4 * R, np.float32))
    assert not c0_ref.any() and not h0_ref.any()                       # KAT: zero weights -> zero state
    i_, j_, f_, o_ = np.split(s + b, 4, axis=1)
    c_fs = ao.sigmoid(i_) * np.tanh(j_)
    h_fs = np.tanh(c_fs) * ao.sigmoid(o_)
    slabs_d, b_d = t(slabs), t(b)
    H.check(lib.air_lstm_first_step(_p(slabs_d), 5, _p(b_d), _p(acts), _p(c1), _p(h1), None, Bn, R, _stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(c1.cpu().numpy(), c_fs, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), h_fs, rtol=2e-6, atol=2e-6)
    # backward of the cell vs torch autograd of the same formulas
    pre_t, c_t = torch.tensor(pre, dtype=torch.float64, requires_grad=True), torch.tensor(c, dtype=torch.float64, requires_grad=True)
    ii, jj, ff, oo = torch.split(pre_t, R, dim=1)
    cn = c_t * torch.sigmoid(ff + 1.0) + torch.sigmoid(ii) * torch.tanh(jj)
    hn = torch.tanh(cn) * torch.sigmoid(oo)
    dh, dc = rng.randn(Bn, R).astype(np.float32), rng.randn(Bn, R).astype(np.float32)
    (hn * torch.tensor(dh, dtype=torch.float64) + cn * torch.tensor(dc, dtype=torch.float64)).sum().backward()
    H.check(lib.air_lstm_gates_fwd(_p(pre_d), _p(c_d), _p(acts), _p(c1), _p(h1), Bn, R, _stream()))
    dh_d, dc_d = t(dh), t(dc)
    dg, dcp = torch.zeros(Bn, 4 * R, device=dev), torch.zeros(Bn, R, device=dev)
    H.check(lib.air_lstm_gates_bwd(_p(dh_d), _p(dc_d), _p(acts), _p(c_d), _p(c1), _p(dg), _p(dcp), None, 0, Bn, R, _stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dg.cpu().numpy(), pre_t.grad.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dcp.cpu().numpy(), c_t.grad.numpy(), rtol=2e-5, atol=2e-6)
    # re-parameterisation: z = mean + eps * sqrt(exp(log_var))   (vae.py:22-24)
    ml = (rng.randn(Bn, 2 * Z) * 0.5).astype(np.float32)
    eps = rng.randn(Bn, Z).astype(np.float32)
    z_ref = ml[:, :Z] + eps * np.sqrt(np.exp(ml[:, Z:]))
    ml_d, eps_d, zs = t(ml), t(eps), torch.zeros(Bn, Z, device=dev)
    H.check(lib.air_reparam_fwd(_p(ml_d), _p(eps_d), _p(zs), Bn, Z, _stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(zs.cpu().numpy(), z_ref, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("sched", [
    dict(init=10000.0, iters=3000, factor=0.1, min=1e-9, staircase=False, log=True),      # training.py:110-115
    dict(init=1.0, iters=1000, factor=0.5, staircase=True),                                # staircase, no clamp, no log
    dict(init=2.0, iters=500, factor=3.0, max=40.0, staircase=False),                      # growing, clamped from above
    dict(init=5.0, iters=700, factor=0.2, min=0.3, max=2.0, staircase=True, log=True),     # everything at once
])
def test_annealing_schedule_variants_match_oracle(H, sched):
    """_create_annealed_tensor (air_model.py:94-121): staircase / min / max / log in every combination,
    evaluated on the device at several global steps vs oracle.annealed_value."""
    dev = "cuda"
    flags = (H.SCHED_STAIRCASE if sched.get("staircase") else 0) | (H.SCHED_HAS_MIN if "min" in sched else 0) | \
            (H.SCHED_HAS_MAX if "max" in sched else 0) | (H.SCHED_LOG if sched.get("log") else 0)
    arr = np.zeros(1, dtype=[("slot", "<i4"), ("flags", "<i4"), ("init", "<f4"), ("iters", "<f4"),
                             ("factor", "<f4"), ("vmin", "<f4"), ("vmax", "<f4")])
    arr[0] = (H.DYN_TEMPERATURE, flags, sched["init"], sched["iters"], sched["factor"], sched.get("min", 0.0), sched.get("max", 0.0))
    tab = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
    dyn = torch.zeros(H.DYN_COUNT, device=dev)
    for step in (0, 1, 499, 500, 999, 1000, 2500, 12345, 40000):
        ist = torch.tensor([step, 0, 0, 0], dtype=torch.int32, device=dev)
        H.check(H.lib().air_step_begin(_p(tab), 1, _p(dyn), _p(ist), None, 0, None, 0, C.c_uint64(0), None, None, 0, _stream()))
        torch.cuda.synchronize()
        ref = float(ao.annealed_value(sched, step))
        got = float(dyn[H.DYN_TEMPERATURE])
        assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (step, got, ref)


def _softplus_tf(x):
    x = np.asarray(x, np.float64)
    return np.where(x > 13.942384719848633, x, np.where(x < -13.942384719848633, np.exp(x), np.log1p(np.exp(x))))


@pytest.mark.parametrize("M,Hd,Z", [(192, 256, 50), (37, 192, 50), (16, 256, 64), (5, 64, 2)])
def test_vae_bottleneck_forward_matches_the_formulas(H, M, Hd, Z):
    """air_vae_bottleneck_fwd = vae.py:16-30 in one launch: mean | log-variance, the reparameterised
    sample and the first generative layer.  Reference: float64 on bf16-rounded GEMM operands."""
    rng = np.random.RandomState(3)
    K1 = 256
    X = rng.uniform(0, 2, (M, K1)).astype(np.float32)
    Wml = rng.uniform(-0.1, 0.1, (K1, 2 * Z)).astype(np.float32)
    bml = rng.uniform(-0.1, 0.1, 2 * Z).astype(np.float32)
    eps = rng.randn(M, Z).astype(np.float32)
    Wg = rng.uniform(-0.3, 0.3, (Z, Hd)).astype(np.float32)
    bg = rng.uniform(-0.1, 0.1, Hd).astype(np.float32)
    t = {k: torch.tensor(v, device="cuda") for k, v in dict(X=X, Wml=Wml, bml=bml, eps=eps, Wg=Wg, bg=bg).items()}
    ml = torch.full((M, 2 * Z), float("nan"), device="cuda")
    z = torch.full((M, Z), float("nan"), device="cuda")
    g = torch.full((M, Hd), float("nan"), device="cuda")
    a = H.BottleneckFwd(_p(t["X"]), _p(t["Wml"]), _p(t["bml"]), _p(t["eps"]), _p(t["Wg"]), _p(t["bg"]), _p(ml), _p(z), _p(g),
                        M, K1, Z, Hd, K1)
    H.check(H.lib().air_vae_bottleneck_fwd(C.byref(a), _stream()))
    torch.cuda.synchronize()
    ml_ref = _bf16_round(X).astype(np.float64) @ _bf16_round(Wml).astype(np.float64) + bml
    z_ref = ml_ref[:, :Z] + eps * np.sqrt(np.exp(ml_ref[:, Z:]))
    assert np.abs(ml.cpu().numpy() - ml_ref).max() < 2e-5
    assert np.abs(z.cpu().numpy() - z_ref).max() < 2e-5
    # the second product sees the sample the kernel made (rounded to bf16 as the MFMA operand is)
    g_ref = _softplus_tf(_bf16_round(z.cpu().numpy()).astype(np.float64) @ _bf16_round(Wg).astype(np.float64) + bg)
    assert np.abs(g.cpu().numpy() - g_ref).max() < 2e-5
    # and equals the two-launch path it replaces
    ml2, z2, g2 = torch.empty_like(ml), torch.empty_like(z), torch.empty_like(g)
    g1 = _gemm_struct(H, t["X"], t["Wml"], ml2, M, 2 * Z, K1, K1, 2 * Z, 2 * Z, 1, bias=t["bml"], epi=H.EPI_REPARAM_FWD,
                      p0=t["eps"], q0=z2)
    H.check(H.lib().air_gemm(C.byref(g1), _stream()))
    g2s = _gemm_struct(H, z2, t["Wg"], g2, M, Hd, Z, Z, Hd, Hd, 1, bias=t["bg"], act=H.ACT_SOFTPLUS)
    H.check(H.lib().air_gemm(C.byref(g2s), _stream()))
    torch.cuda.synchronize()
    assert (ml - ml2).abs().max() < 1e-5 and (z - z2).abs().max() < 1e-5 and (g - g2).abs().max() < 2e-5
    # the three operands from their bf16 twins (X16 / Wml16 / Wg16): the same bits in every output
    ml3, z3, g3 = torch.full_like(ml, float("nan")), torch.full_like(z, float("nan")), torch.full_like(g, float("nan"))
    tw = [_bf16_twin(H, t[k]) for k in ("X", "Wml", "Wg")]
    a3 = H.BottleneckFwd(_p(t["X"]), _p(t["Wml"]), _p(t["bml"]), _p(t["eps"]), _p(t["Wg"]), _p(t["bg"]), _p(ml3), _p(z3), _p(g3),
                         M, K1, Z, Hd, K1, None, None, _p(tw[0]), _p(tw[1]), _p(tw[2]))
    H.check(H.lib().air_vae_bottleneck_fwd(C.byref(a3), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(ml, ml3) and torch.equal(z, z3) and torch.equal(g, g3)


@pytest.mark.parametrize("M,K1,Z", [(192, 256, 50), (37, 200, 50), (16, 64, 64), (5, 256, 2)])
def test_vae_bottleneck_backward_matches_the_formulas(H, M, K1, Z):
    """air_vae_bottleneck_bwd: d_z = dG.Wg^T, the reparameterisation + KL gradients (vae.py:22-24,
    air_model.py:386-392, masked by the step's stopping mask), d_x = (d_ml.Wml^T) * softplus'(x)."""
    rng = np.random.RandomState(4)
    Hd = 256
    dG = rng.randn(M, Hd).astype(np.float32) * 0.1
    Wg = rng.uniform(-0.3, 0.3, (Z, Hd)).astype(np.float32)
    ml = rng.uniform(-1, 1, (M, 2 * Z)).astype(np.float32)
    eps = rng.randn(M, Z).astype(np.float32)
    att = np.zeros((M, H.ATT_STRIDE), np.float32)
    att[:, H.ATT_MASK] = rng.randint(0, 2, M)
    dyn = np.zeros(32, np.float32)
    dyn[H.DYN_GRAD_SCALE], dyn[H.DYN_VAE_PV], dyn[H.DYN_VAE_PM] = 1.0 / 64, 0.8, 0.1
    Wml = rng.uniform(-0.1, 0.1, (K1, 2 * Z)).astype(np.float32)
    x = rng.uniform(0.01, 2, (M, K1)).astype(np.float32)
    t = {k: torch.tensor(v, device="cuda") for k, v in dict(dG=dG, Wg=Wg, ml=ml, eps=eps, att=att, dyn=dyn, Wml=Wml, x=x).items()}
    d_ml = torch.full((M, 2 * Z), float("nan"), device="cuda")
    d_x = torch.full((M, K1), float("nan"), device="cuda")
    a = H.BottleneckBwd(_p(t["dG"]), _p(t["Wg"]), _p(t["ml"]), _p(t["eps"]), _p(t["att"]), _p(t["dyn"]), _p(t["Wml"]), _p(t["x"]),
                        _p(d_ml), _p(d_x), M, K1, Z, Hd)
    H.check(H.lib().air_vae_bottleneck_bwd(C.byref(a), _stream()))
    torch.cuda.synchronize()
    dz = _bf16_round(dG).astype(np.float64) @ _bf16_round(Wg).astype(np.float64).T
    klg = att[:, H.ATT_MASK:H.ATT_MASK + 1].astype(np.float64) * dyn[H.DYN_GRAD_SCALE]
    var = np.exp(ml[:, Z:].astype(np.float64))
    dmean = dz + klg * (ml[:, :Z] - dyn[H.DYN_VAE_PM]) / dyn[H.DYN_VAE_PV]
    dlv = dz * eps * 0.5 * np.sqrt(var) + klg * 0.5 * (var / dyn[H.DYN_VAE_PV] - 1.0)
    ref = np.concatenate([dmean, dlv], 1)
    got = d_ml.cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5
    dx_ref = (_bf16_round(got).astype(np.float64) @ _bf16_round(Wml).astype(np.float64).T) * (1.0 - np.exp(-x.astype(np.float64)))
    assert np.abs(d_x.cpu().numpy() - dx_ref).max() < 2e-5
    # the two-launch path it replaces
    d_ml2, d_x2 = torch.empty_like(d_ml), torch.empty_like(d_x)
    g1 = _gemm_struct(H, t["dG"], t["Wg"], d_ml2, M, Z, Hd, Hd, Hd, 2 * Z, 1, transB=1, epi=H.EPI_REPARAM_BWD,
                      p0=t["ml"], p1=t["eps"], p2=t["att"], p3=t["dyn"])
    H.check(H.lib().air_gemm(C.byref(g1), _stream()))
    g2 = _gemm_struct(H, d_ml2, t["Wml"], d_x2, M, K1, 2 * Z, 2 * Z, 2 * Z, K1, 1, transB=1, aux=t["x"], ldaux=K1,
                      actgrad=H.GRAD_SOFTPLUS)
    H.check(H.lib().air_gemm(C.byref(g2), _stream()))
    torch.cuda.synchronize()
    # (d_ml differs in the last fp32 bits between the two accumulation orders; where that flips its bf16
    # rounding as the second product's operand, d_x moves by one bf16 ulp of d_ml times a weight)
    assert (d_ml - d_ml2).abs().max() < 1e-5 and (d_x - d_x2).abs().max() < 3e-4
    # the three operands from their bf16 twins (dG16 / Wg16 / Wml16): the same bits
    d_ml3, d_x3 = torch.full_like(d_ml, float("nan")), torch.full_like(d_x, float("nan"))
    tw = [_bf16_twin(H, t[k]) for k in ("dG", "Wg", "Wml")]
    a3 = H.BottleneckBwd(_p(t["dG"]), _p(t["Wg"]), _p(t["ml"]), _p(t["eps"]), _p(t["att"]), _p(t["dyn"]), _p(t["Wml"]), _p(t["x"]),
                         _p(d_ml3), _p(d_x3), M, K1, Z, Hd, None, None, _p(tw[0]), _p(tw[1]), _p(tw[2]))
    H.check(H.lib().air_vae_bottleneck_bwd(C.byref(a3), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(d_ml, d_ml3) and torch.equal(d_x, d_x3)


def test_vae_bottleneck_limits(H):
    a = H.BottleneckFwd()
    buf = torch.zeros(1024, device="cuda")
    for f in ("X", "Wml", "bml", "eps", "Wg", "bg", "ml", "z", "g"):
        setattr(a, f, buf.data_ptr())
    a.M, a.K1, a.Z, a.H, a.ldx = 4, 128, 50, 256, 128
    assert H.lib().air_vae_bottleneck_fwd(C.byref(a), _stream()) == -2          # K1 != 256: callers fall back
    a.K1, a.ldx, a.Z = 256, 256, 51
    assert H.lib().air_vae_bottleneck_fwd(C.byref(a), _stream()) == -3          # odd Z


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,R,K", [(64, 256, 2500), (37, 12, 300), (16, 4, 64)])
def test_gemm_first_lstm_step_epilogue_matches_product_plus_pointwise_step(H, prec, M, R, K):
    """AIR_EPI_LSTM_FWD0: the hoisted x.Wx and, from zero state, the first BasicLSTMCell step in one launch
    (air_model.py:286, :540) == air_gemm (plain) followed by air_lstm_first_step; the gates against the
    float64 formulas on the operands as the MFMAs see them."""
    rng = np.random.RandomState(11)
    X = rng.uniform(0, 1, (M, K)).astype(np.float32)
    W = rng.uniform(-0.05, 0.05, (K, 4 * R)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, 4 * R).astype(np.float32)
    Xt, Wt, bt = (torch.tensor(v, device="cuda") for v in (X, W, b))
    xw = torch.full((M, 4 * R), float("nan"), device="cuda")
    acts = torch.full((M, 4 * R), float("nan"), device="cuda")
    c, h = torch.full((M, R), float("nan"), device="cuda"), torch.full((M, R), float("nan"), device="cuda")
    g = _gemm_struct(H, Xt, Wt, xw, M, 4 * R, K, K, 4 * R, 4 * R, prec, bias=bt, epi=H.EPI_LSTM_FWD0, q0=acts, q1=c, q2=h)
    H.check(H.lib().air_gemm(C.byref(g), _stream()))
    torch.cuda.synchronize()
    rd = (lambda v: _bf16_round(v)) if prec else (lambda v: np.asarray(v, np.float64))
    ref = rd(X) @ rd(W)
    assert np.abs(xw.cpu().numpy() - ref).max() < (2e-6 if prec == 0 else 2e-5) * np.sqrt(K)
    pre = xw.cpu().numpy().astype(np.float64) + b
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    si, tj, sf, so = sig(pre[:, :R]), np.tanh(pre[:, R:2 * R]), sig(pre[:, 2 * R:3 * R] + 1.0), sig(pre[:, 3 * R:])
    np.testing.assert_allclose(acts.cpu().numpy(), np.concatenate([si, tj, sf, so], 1), atol=2e-6)
    np.testing.assert_allclose(c.cpu().numpy(), si * tj, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), np.tanh(si * tj) * so, atol=2e-6)
    # the two launches it replaces, fed the same product: bit-identical gates
    acts2, c2, h2 = torch.empty_like(acts), torch.empty_like(c), torch.empty_like(h)
    H.check(H.lib().air_lstm_first_step(_p(xw), 1, _p(bt), _p(acts2), _p(c2), _p(h2), None, M, R, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(acts, acts2) and torch.equal(c, c2) and torch.equal(h, h2)


# ---- graph-order sampler backward at every canvas regime (small: all taps resident; large: per-tap staging) ----

def _write_theta(s, x, y):
    """theta_recon of air_model.py:353-356 in the kernel's fp32 op order: [[1/s, 0, -x/s], [0, 1/s, -y/s]]."""
    one = np.float32(1.0)
    th = np.zeros((len(s), 2, 3), np.float32)
    th[:, 0, 0] = th[:, 1, 1] = one / s
    th[:, 0, 2], th[:, 1, 2] = (-x) / s, (-y) / s
    return th


@pytest.mark.parametrize("literal,order", [(2, "sequential"), (4, "carried16")])
@pytest.mark.parametrize("Cc,w,N,B", [(50, 28, 3, 6), (71, 28, 2, 5), (128, 28, 5, 4), (128, 20, 2, 3), (50, 32, 2, 3), (40, 9, 2, 4)])
def test_write_bwd_graph_order_matches_oracle_all_canvas_sizes(H, Cc, w, N, B, literal, order):
    """air_write_bwd(literal=2) -- backward="reference" -- against oracle.transformer_backward
    (pinned to the reference's executed graph, transformer.py:56-117 under tf.gradients) on random inputs, and
    air_write_bwd(literal=4) -- backward="reference_carried" -- against the same function with order="carried16" (short
    streams as the reference, the chunks of long ones walked from a carried stand-in for the reference's accumulator,
    oracle.carried_segment_sum).
    C = 50 takes write_bwd_{graph,carried}_kernel<true> (all four taps' terms resident), C >= 63 the per-tap staged
    <false> variant that every large canvas (BASELINE configs[3]: 128x128, N = 5) runs.  d_gen_pre (the
    UnsortedSegmentSum result times SigmoidGrad) BIT FOR BIT, residue included; theta / z legs <= 2e-5."""
    name = C.create_string_buffer(96)
    rng = np.random.RandomState(Cc * 3 + w + N)
    s = rng.uniform(0.12, 0.6, (N, B)).astype(np.float32)
    x = rng.uniform(-0.8, 0.8, (N, B)).astype(np.float32)
    y = rng.uniform(-0.8, 0.8, (N, B)).astype(np.float32)
    s[0, 0], x[0, 0], y[0, 0] = 0.3, -0.95, 0.9            # glimpse in a canvas corner: one corner slot owns most pixels
    s[0, 1], x[0, 1], y[0, 1] = 0.97, 0.02, -0.01          # the glimpse covers the canvas: no out-of-range pixels, long interior streams
    z = rng.uniform(0.05, 1.0, (N, B)).astype(np.float32)
    mask = np.ones((N, B), np.float32)
    mask[N - 1, B - 1] = 0.0                              # a stopped item: Select(active, ., 0)
    att = np.zeros((N, B, H.ATT_STRIDE), np.float32)
    att[:, :, H.ATT_S], att[:, :, H.ATT_X], att[:, :, H.ATT_Y], att[:, :, H.ATT_Z] = s, x, y, z
    att[:, :, H.ATT_MASK] = mask
    # d loss / d reconstruction with the poles of the Bernoulli ELBO (1e9 / B at unexplained ink)
    g = (rng.randn(B, Cc * Cc) * np.where(rng.uniform(size=(B, Cc * Cc)) < 0.08, 1e7, 1e-2)).astype(np.float32)
    vrec = rng.uniform(0.01, 0.99, (N, B, w * w)).astype(np.float32)
    att_d, g_d, v_d = (torch.tensor(v, device="cuda") for v in (att, g, vrec))
    dgen = torch.full((N, B, w * w), 7.0, device="cuda")
    dsx = torch.full((N, B, 4), 7.0, device="cuda")
    wb = H.WriteBwd(_p(g_d), _p(v_d), _p(att_d), _p(dgen), _p(dsx), B, N, Cc, w, literal, None, None, None, None)
    H.check(H.lib().air_write_bwd_kernel_name(C.byref(wb), name, 96))
    assert name.value.decode() == "write_bwd_%s_kernel<%s>" % ({2: "graph", 4: "carried"}[literal], "true" if Cc <= 62 else "false")
    H.check(H.lib().air_write_bwd(C.byref(wb), _stream()), "air_write_bwd")
    torch.cuda.synchronize()
    dgen, dsx = dgen.cpu().numpy(), dsx.cpu().numpy()
    residue = 0.0
    for t in range(N):
        th = _write_theta(s[t], x[t], y[t])
        U = vrec[t].reshape(B, w, w)
        d_out = (z[t][:, None] * g).reshape(B, Cc, Cc)                        # canvas/mul_grad: z * Select_grad
        dU, dth = ao.transformer_backward(U, th, (Cc, Cc), d_out, order=order)
        ref = ((dU.reshape(B, -1) * vrec[t]) * (np.float32(1.0) - vrec[t])).astype(np.float32)   # SigmoidGrad
        patch = ao.transformer(U, th, (Cc, Cc)).reshape(B, -1).astype(np.float64)
        for b in range(B):
            if mask[t, b] == 0.0:
                assert not dgen[t, b].any() and not dsx[t, b].any()
                continue
            assert np.array_equal(dgen[t, b], ref[b]), (t, b, float(np.abs(dgen[t, b] - ref[b]).max()))
            residue = max(residue, float(np.abs(ref[b]).max()))
            sb, xb, yb = np.float64(s[t, b]), np.float64(x[t, b]), np.float64(y[t, b])
            d00, d02, d11, d12 = (np.float64(dth[b, 0, 0]), np.float64(dth[b, 0, 2]),
                                  np.float64(dth[b, 1, 1]), np.float64(dth[b, 1, 2]))
            want = np.array([-(d00 + d11) / sb ** 2 + (d02 * xb + d12 * yb) / sb ** 2, -d02 / sb, -d12 / sb,
                             float((g[b].astype(np.float64) * patch[b]).sum())])
            terms = np.array([abs(d00) / sb ** 2 + abs(d11) / sb ** 2 + abs(d02 * xb) / sb ** 2 + abs(d12 * yb) / sb ** 2,
                              abs(d02) / sb, abs(d12) / sb, float(np.abs(g[b].astype(np.float64) * patch[b]).sum())])
            # fixed-order block reductions over C*C pixels vs numpy's matmul: relative to the summed magnitudes
            assert (np.abs(dsx[t, b] - want) <= 2e-5 * np.maximum(terms, 1e-30) + 2e-5 * np.abs(want)).all(), \
                (t, b, dsx[t, b], want)
    assert residue > 1.0          # the out-of-range residue is present in what was compared


@pytest.mark.parametrize("Cc,w", [(50, 28), (128, 28), (97, 20)])
def test_attend_bwd_graph_order_read_gradient_matches_oracle(H, Cc, w):
    """air_attend_bwd(literal=2) at small and large canvases (C = 128 takes the bounding-box staging): the
    read transformer's gradient wrt theta = [[s,0,x],[0,s,y]] (air_model.py:324-333) from
    oracle.transformer_backward in the graph's op order, pushed through the sampling (sigmoid / tanh of
    mean + eps*sd, :300-303, :317-320), the Gaussian KLs and the Concrete z_pres (concrete.py:20-43) in fp64."""
    rng = np.random.RandomState(Cc + w)
    N, B, Hs, Hh, Hz = 2, 5, 64, 64, 64
    HT = 2 * Hs + 2 * Hh + Hz
    canvas = rng.uniform(0, 1, (B, Cc * Cc)).astype(np.float32)
    o7 = (rng.randn(N, B, H.OUT_STRIDE) * 0.5).astype(np.float32)
    e_s, e_h = rng.randn(N, B, 1).astype(np.float32), rng.randn(N, B, 2).astype(np.float32)
    sd = lambda lv: np.sqrt(np.exp(lv.astype(np.float64)))  # noqa: E731
    s = 1.0 / (1.0 + np.exp(-(o7[..., 0] + e_s[..., 0] * sd(o7[..., 1]))))
    xs = np.tanh(o7[..., 2] + e_h[..., 0] * sd(o7[..., 4]))
    ys = np.tanh(o7[..., 3] + e_h[..., 1] * sd(o7[..., 5]))
    s, xs, ys = (v.astype(np.float32) for v in (s, xs, ys))
    u = rng.uniform(0.05, 0.95, (N, B))
    T, plo, gsc = 1.0, -2.0, 1.0 / 64
    ypre = ((o7[..., 6] + np.log(u) - np.log(1 - u)) / T).astype(np.float32)
    zz = (1.0 / (1.0 + np.exp(-ypre.astype(np.float64)))).astype(np.float32)
    att = np.zeros((N, B, H.ATT_STRIDE), np.float32)
    att[..., H.ATT_S], att[..., H.ATT_X], att[..., H.ATT_Y] = s, xs, ys
    att[..., H.ATT_Z], att[..., H.ATT_ZPRE] = zz, ypre
    att[..., H.ATT_MASK], att[..., H.ATT_MASK_PREV] = 1.0, 1.0
    att[1, 0, H.ATT_MASK] = 0.0
    d_win = rng.randn(N, B, w * w).astype(np.float32)
    d_sxyw = rng.randn(N, B, 4).astype(np.float32)
    dyn = np.zeros(H.DYN_COUNT, np.float32)
    dyn[H.DYN_PRIOR_LOG_ODDS], dyn[H.DYN_TEMPERATURE], dyn[H.DYN_STOP_THRESHOLD] = plo, T, 0.99
    dyn[H.DYN_SCALE_PM], dyn[H.DYN_SCALE_PV], dyn[H.DYN_SHIFT_PM], dyn[H.DYN_SHIFT_PV] = -1.0, 0.1, 0.0, 1.0
    dyn[H.DYN_VAE_PM], dyn[H.DYN_VAE_PV], dyn[H.DYN_GRAD_SCALE] = 0.0, 1.0, gsc
    t_ = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    hid, wout = torch.ones(N, B, HT, device="cuda"), torch.zeros(7, 64, device="cuda")
    d_hid, d_out7 = torch.zeros(N, B, HT, device="cuda"), torch.zeros(N, B, H.OUT_STRIDE, device="cuda")
    cv, es_d, eh_d, dyn_d, o7_d, att_d, dw_d, ds_d = (t_(v) for v in (canvas, e_s, e_h, dyn, o7, att, d_win, d_sxyw))
    ab = H.AttendBwd(_p(hid), _p(wout), _p(cv), _p(es_d), _p(eh_d), _p(dyn_d), _p(o7_d), _p(att_d),
                     _p(dw_d), _p(ds_d), _p(d_hid), _p(d_out7), B, N, Cc, w, Hs, Hh, Hz, 64, 2)
    H.check(H.lib().air_attend_bwd(C.byref(ab), _stream()), "air_attend_bwd")
    torch.cuda.synchronize()
    got = d_out7.cpu().numpy().astype(np.float64)
    for t in range(N):
        th = np.zeros((B, 2, 3), np.float32)
        th[:, 0, 0] = th[:, 1, 1] = s[t]
        th[:, 0, 2], th[:, 1, 2] = xs[t], ys[t]
        _, dth = ao.transformer_backward(canvas.reshape(B, Cc, Cc), th, (w, w), d_win[t].reshape(B, w, w))
        dth = dth.astype(np.float64)
        d_s = dth[:, 0, 0] + dth[:, 1, 1] + d_sxyw[t, :, 0]
        d_x, d_y, d_z = dth[:, 0, 2] + d_sxyw[t, :, 1], dth[:, 1, 2] + d_sxyw[t, :, 2], d_sxyw[t, :, 3].astype(np.float64)
        S, X, Y = (v[t].astype(np.float64) for v in (s, xs, ys))
        klg = att[t, :, H.ATT_MASK].astype(np.float64) * gsc
        da_s, da_x, da_y = d_s * S * (1 - S), d_x * (1 - X * X), d_y * (1 - Y * Y)
        o = o7[t].astype(np.float64)
        sds, sdx, sdy = sd(o7[t, :, 1]), sd(o7[t, :, 4]), sd(o7[t, :, 5])
        ref = np.zeros((B, 7))
        ref[:, 0] = da_s + klg * (o[:, 0] + 1.0) / 0.1
        ref[:, 1] = da_s * e_s[t, :, 0] * 0.5 * sds + klg * 0.5 * (sds * sds / 0.1 - 1.0)
        ref[:, 2] = da_x + klg * o[:, 2]
        ref[:, 3] = da_y + klg * o[:, 3]
        ref[:, 4] = da_x * e_h[t, :, 0] * 0.5 * sdx + klg * 0.5 * (sdx * sdx - 1.0)
        ref[:, 5] = da_y * e_h[t, :, 1] * 0.5 * sdy + klg * 0.5 * (sdy * sdy - 1.0)
        yp, Z_ = ypre[t].astype(np.float64), zz[t].astype(np.float64)
        eq, ep = np.exp(-yp * T + o[:, 6]), np.exp(-yp * T + plo)
        rq, rp = eq / (1 + eq + 1e-9), ep / (1 + ep + 1e-9)
        dkl = att[t, :, H.ATT_MASK_PREV].astype(np.float64) * gsc
        ref[:, 6] = (d_z * Z_ * (1 - Z_) + dkl * 2 * T * (rq - rp)) / T + dkl * (1 - 2 * rq)
        for k in range(7):
            scale = max(np.abs(ref[:, k]).max(), 1e-6)
            assert np.abs(got[t, :, k] - ref[:, k]).max() <= 1e-4 * scale, (t, k, got[t, :, k], ref[:, k])


# ---- bf16 twins: operands read as bf16 (written by the producer / by Adam) -- bit-identical to rounding on the way into LDS ----

def _bf16_twin(H, t):
    """RNE bf16 twin of a device fp32 tensor through the ABI's own converter (as int16 storage)."""
    tw = torch.empty(t.shape, dtype=torch.int16, device=t.device)
    H.check(H.lib().air_bf16_twin(_p(t), _p(tw), t.numel(), _stream()))
    return tw


def _kernel_name(H, g):
    buf = C.create_string_buffer(128)
    H.check(H.lib().air_gemm_kernel_name(C.byref(g), buf, 128))
    return buf.value.decode()


def test_bf16_twin_converter_is_rne(H):
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.randn(4099).astype(np.float32) * 10.0 ** rng.randint(-6, 6, 4099),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, -3.0e38, 1e-40], np.float32)]).astype(np.float32)
    xt = torch.tensor(x, device="cuda")
    tw = _bf16_twin(H, xt)
    torch.cuda.synchronize()
    assert torch.equal(tw.view(torch.bfloat16), xt.to(torch.bfloat16))


TWIN_GEMMS = [
    # (M, N, K, transB, tile) -- the train step's shapes (Cfg-A and the 128x128 stress batch) + multi-round depths
    (192, 320, 256, 0, (0, 0)), (192, 512, 784, 0, (0, 0)), (192, 256, 512, 0, (0, 0)), (192, 784, 512, 0, (0, 0)),
    (192, 512, 784, 1, (0, 0)), (192, 256, 512, 1, (0, 0)), (192, 784, 512, 1, (0, 0)), (192, 256, 320, 1, (0, 0)),
    (64, 256, 1024, 1, (1, 1)), (1280, 512, 784, 0, (0, 0)), (1280, 784, 512, 1, (0, 0)), (1280, 512, 256, 1, (2, 2)),
    (50, 96, 2048, 0, (1, 1)), (50, 96, 2048, 1, (1, 1)), (37, 200, 1160, 0, (2, 2)),
]


@pytest.mark.parametrize("M,N,K,tb,tile", TWIN_GEMMS)
def test_gemm_bf16_twins_bit_identical(H, M, N, K, tb, tile):
    """air_gemm(precision=1) with bf16 twins of both operands (air_gemm_t.A16 / B16) against the same call
    without them (fp32 operands rounded on their way into LDS): the same RNE rounding, the same k order
    and reduction, so C is BIT-IDENTICAL; the twin C16 the epilogue writes is bf16(C)."""
    rng = np.random.RandomState(M + N + K + tb)
    A = torch.tensor(rng.uniform(-1, 1, (M, K)).astype(np.float32), device="cuda")
    B = torch.tensor(rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32), device="cuda")
    bias = torch.tensor(rng.uniform(-1, 1, N).astype(np.float32), device="cuda")
    aux = torch.tensor(rng.uniform(0.1, 2, (M, N)).astype(np.float32), device="cuda")
    A16, B16 = _bf16_twin(H, A), _bf16_twin(H, B)
    outs = []
    for twins in (False, True):
        Ct = torch.full((M, N), float("nan"), device="cuda")
        C16 = torch.zeros((M, N), dtype=torch.int16, device="cuda")
        kw = dict(transB=tb, tile_m=tile[0], tile_n=tile[1], bias=bias, act=H.ACT_SOFTPLUS, C16=C16)
        if tb:
            kw = dict(transB=1, tile_m=tile[0], tile_n=tile[1], aux=aux, ldaux=N, actgrad=H.GRAD_SOFTPLUS, C16=C16)
        if twins:
            kw.update(A16=A16, B16=B16)
        g = _gemm_struct(H, A, B, Ct, M, N, K, K, B.shape[1], N, 1, **kw)
        name = _kernel_name(H, g)
        assert name.startswith("gemm_bf16tw_kernel" if twins else "gemm_bf16v2_kernel"), name
        H.check(H.lib().air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        assert torch.equal(C16.view(torch.bfloat16), Ct.to(torch.bfloat16))
        outs.append(Ct)
    assert torch.equal(outs[0], outs[1])
    ref = _ref_gemm(A.cpu().numpy(), B.cpu().numpy(), 0, tb, 1)
    if not tb:
        ref = np.log1p(np.exp(ref + bias.cpu().numpy().astype(np.float64)))
    else:
        ref = ref * (1.0 - np.exp(-aux.cpu().numpy().astype(np.float64)))
    assert np.abs(outs[1].cpu().numpy() - ref).max() / np.sqrt(K) < 2e-5


def test_gemm_bf16_twins_fp32_a_split_k_bit_identical(H):
    """The hoisted x.Wx: fp32 image batch (the caller's tensor) x bf16 shadow of Wx, split-K slabs."""
    rng = np.random.RandomState(12)
    for M, N, K, ks, tile in ((64, 1024, 2500, 4, (2, 2)), (256, 1024, 16384, 4, (4, 2))):
        A = torch.tensor(rng.uniform(0, 1, (M, K)).astype(np.float32), device="cuda")
        B = torch.tensor(rng.uniform(-0.05, 0.05, (K, N)).astype(np.float32), device="cuda")
        B16 = _bf16_twin(H, B)
        S = H.lib().air_gemm_slabs(K, ks)
        outs = []
        for twins in (False, True):
            Ct = torch.full((S, M, N), float("nan"), device="cuda")
            kw = dict(ksplit=ks, tile_m=tile[0], tile_n=tile[1])
            if twins:
                kw["B16"] = B16
            g = _gemm_struct(H, A, B, Ct, M, N, K, K, N, N, 1, **kw)
            assert _kernel_name(H, g).startswith("gemm_bf16tw_kernel" if twins else "gemm_bf16v2_kernel")
            H.check(H.lib().air_gemm(C.byref(g), _stream()))
            torch.cuda.synchronize()
            outs.append(Ct)
        assert torch.equal(outs[0], outs[1]), (M, N, K)


def test_gemm_bf16_twins_fused_lstm_epilogues_bit_identical(H):
    """AIR_EPI_LSTM_FWD / LSTM_BWD / LSTM_BWD_TAIL on twin operands: every output bit-identical to the
    fp32-operand launch, and the twins they write (h, dgates, the final running sum) are bf16 of the fp32 outputs."""
    dev, lib = "cuda", H.lib()
    rng = np.random.RandomState(13)
    Bn, R, HT = 64, 256, 320
    f = lambda *s: torch.tensor(rng.uniform(-1, 1, s).astype(np.float32), device=dev)  # noqa: E731
    i16 = lambda *s: torch.zeros(*s, dtype=torch.int16, device=dev)  # noqa: E731
    h, Wh, bias, c_prev, slabs = f(Bn, R), f(R, 4 * R) * 0.1, f(4 * R) * 0.1, f(Bn, R), f(4, Bn, 4 * R) * 0.3
    res = []
    for twins in (False, True):
        acts, c1, h1, dummy, h16 = f(Bn, 4 * R), f(Bn, R), f(Bn, R), f(Bn, 4 * R), i16(Bn, R)
        kw = dict(bias=bias, addend=slabs, ldadd=4 * R, addend_slabs=4, epi=H.EPI_LSTM_FWD, p0=c_prev, q0=acts, q1=c1, q2=h1, q2_16=h16)
        if twins:
            kw.update(A16=_bf16_twin(H, h), B16=_bf16_twin(H, Wh))
        g = _gemm_struct(H, h, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, 1, **kw)
        assert ("tw_kernel" in _kernel_name(H, g)) == twins
        H.check(lib.air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        assert torch.equal(h16.view(torch.bfloat16), h1.to(torch.bfloat16))
        res.append((acts, c1, h1))
    for a0, a1 in zip(*res):
        assert torch.equal(a0, a1)
    acts0, c0 = res[0][0], res[0][1]
    # BPTT step: dgates[t+1] . Wh^T (+ heads' d h) -> LSTM cell backward; running sum accumulated
    dgn, dh_heads, dc_in, ds_init = f(Bn, 4 * R), f(Bn, R), f(Bn, R), f(Bn, 4 * R)
    res = []
    for twins in (False, True):
        dh, dg, dcp, ds = f(Bn, R), f(Bn, 4 * R), f(Bn, R), ds_init.clone()
        dg16, ds16 = i16(Bn, 4 * R), i16(Bn, 4 * R)
        kw = dict(transB=1, addend=dh_heads, ldadd=R, epi=H.EPI_LSTM_BWD, p0=acts0, p1=c_prev, p2=c0, p3=dc_in,
                  q0=dg, q1=dcp, q2=ds, i0=1, q0_16=dg16, q2_16=ds16)
        if twins:
            kw.update(A16=_bf16_twin(H, dgn), B16=_bf16_twin(H, Wh))
        g = _gemm_struct(H, dgn, Wh, dh, Bn, R, 4 * R, 4 * R, 4 * R, R, 1, **kw)
        assert ("tw_kernel" in _kernel_name(H, g)) == twins
        H.check(lib.air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        assert torch.equal(dg16.view(torch.bfloat16), dg.to(torch.bfloat16))
        assert torch.equal(ds16.view(torch.bfloat16), ds.to(torch.bfloat16))
        res.append((dg, dcp, ds))
    for a0, a1 in zip(*res):
        assert torch.equal(a0, a1)
    # heads' d h of all steps, the last step's rows straight through the cell backward
    NB, tl = 3 * Bn, 2 * Bn
    d_hid, Whid = f(NB, HT), f(R, HT) * 0.1
    res = []
    for twins in (False, True):
        dhh, dg, dcp, ds = f(NB, R), f(Bn, 4 * R), f(Bn, R), f(Bn, 4 * R)
        dg16 = i16(Bn, 4 * R)
        kw = dict(transB=1, epi=H.EPI_LSTM_BWD_TAIL, i0=tl, p0=acts0, p1=c_prev, p2=c0, q0=dg, q1=dcp, q2=ds, q0_16=dg16)
        if twins:
            kw.update(A16=_bf16_twin(H, d_hid), B16=_bf16_twin(H, Whid))
        g = _gemm_struct(H, d_hid, Whid, dhh, NB, R, HT, HT, HT, R, 1, **kw)
        assert ("tw_kernel" in _kernel_name(H, g)) == twins
        H.check(lib.air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        assert torch.equal(dg16.view(torch.bfloat16), dg.to(torch.bfloat16))
        res.append((dhh[:tl].clone(), dg, dcp, ds))
    for a0, a1 in zip(*res):
        assert torch.equal(a0, a1)


def test_gemm_ragged_shapes_keep_the_fp32_operand_kernels(H):
    """Twins given but the shape is not made of whole 16-byte bf16 pieces (K % 8, ld % 8, N % 8): the call
    silently takes the fp32-operand kernels -- same results, and it still writes the C16 twin."""
    rng = np.random.RandomState(14)
    for M, N, K, tb in ((40, 100, 50, 0), (40, 50, 100, 1), (33, 64, 36, 0), (33, 36, 64, 1)):
        A = torch.tensor(rng.uniform(-1, 1, (M, K)).astype(np.float32), device="cuda")
        B = torch.tensor(rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32), device="cuda")
        Ct, C16 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, dtype=torch.int16, device="cuda")
        g = _gemm_struct(H, A, B, Ct, M, N, K, K, B.shape[1], N, 1, transB=tb, A16=_bf16_twin(H, A), B16=_bf16_twin(H, B), C16=C16)
        ragged = (K % 8 != 0) or (not tb and N % 8 != 0)
        assert ("tw_kernel" in _kernel_name(H, g)) == (not ragged), (M, N, K, tb)
        H.check(H.lib().air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        assert np.abs(Ct.cpu().numpy() - _ref_gemm(A.cpu().numpy(), B.cpu().numpy(), 0, tb, 1)).max() / np.sqrt(K) < 2e-5
        assert torch.equal(C16.view(torch.bfloat16), Ct.to(torch.bfloat16))


# (problems of >= 512 tiles run in STRIPS with twins, run_strip_bf16: the first case's 2500 x 1024 over K = 64 in masked
# 8-byte pieces, the 8192 x 1024 / 4096 x 2048 ones in 16-byte pieces)
@pytest.mark.parametrize("shapes", WGRAD_CASES + [[(256, 1024, 1280), (2500, 1024, 256)],
                                                  [(256, 1024, 192), (8192, 1024, 256), (100, 36, 64)], [(4096, 2048, 128)],
                                                  # long contractions with 16-byte rows (even / odd / ragged round
                                                  # counts, ragged M and N)
                                                  [(784, 512, 1280), (104, 200, 400), (64, 64, 384), (72, 136, 520)],
                                                  # the gathered-factor contraction of dp_exchange="factors" at world 8:
                                                  # dWx over K = 8 x 64 rows (configs[2]) and 8 x 256 (stress canvases)
                                                  [(2500, 1024, 512)], [(16384, 1024, 2048)]])
def test_wgrad_grouped_bf16_twins_bit_identical(H, shapes):
    """air_wgrad_grouped(precision=1) with bf16 twins of A and dY (air_wgrad_t.A16 / dY16) against the same
    launch without them: dW, db and the global-norm partials BIT-IDENTICAL.  Covers 16-byte rows (ld % 8 == 0),
    8-byte rows (M = 2500, N = 100), problems whose twins are unusable (M = 50, M = 70, N = 33: ld % 4 != 0 --
    they silently take the fp32 operands) and several rounds of images (K = 200, 1280)."""
    dev = "cuda"
    rng = np.random.RandomState(len(shapes))
    outs = []
    ops = []
    for (M, N, K) in shapes:
        At = torch.tensor(rng.randn(K, M).astype(np.float32), device=dev)
        Yt = torch.tensor((rng.randn(K, N) * 10.0 ** rng.randint(-2, 3, (K, 1))).astype(np.float32), device=dev)
        ops.append((At, Yt, _bf16_twin(H, At), _bf16_twin(H, Yt)))
    for twins in (False, True):
        keep, probs = [], []
        for (M, N, K), (At, Yt, A16, Y16) in zip(shapes, ops):
            Wt, bt = torch.full((M, N), float("nan"), device=dev), torch.full((N,), float("nan"), device=dev)
            keep += [Wt, bt]
            probs.append(H.Wgrad(_p(At), _p(Yt), _p(Wt), _p(bt), M, N, K, M, N, N, 0, 0, 0, 0,
                                 _p(A16) if twins else None, _p(Y16) if twins else None))
        arr = (H.Wgrad * len(probs))(*probs)
        nblk = H.lib().air_wgrad_num_blocks(arr, len(probs))
        # strips: only with twins, only for the big problems (>= 512 tiles: 2 column tiles per workgroup, >= 2048: 4)
        saved = 0
        for (M, N, K) in shapes:
            tiles = -(-M // 64) * -(-N // 64)
            if tiles >= 512 and K in (64, 128, 192, 256) and M % 4 == 0 and N % 256 == 0:
                saved += tiles - tiles // (4 if tiles >= 2048 else 2)
        assert H.lib().air_wgrad_num_workgroups(arr, len(probs), 1) == (nblk - saved if twins else nblk)
        assert H.lib().air_wgrad_num_workgroups(arr, len(probs), 0) == nblk
        part = torch.full((nblk,), float("nan"), device=dev)
        ist = torch.zeros(8, dtype=torch.int32, device=dev)
        H.check(H.lib().air_wgrad_grouped(arr, len(probs), 1, _p(part), _p(ist), _stream()), "air_wgrad_grouped")
        torch.cuda.synchronize()
        assert int(ist[H.IST_GLOBAL_STEP]) == 1
        outs.append(keep + [part])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for (M, N, K), (At, Yt, _, _), Wt in zip(shapes, ops, outs[1][0::2]):
        ref = _bf16_round(At.cpu().numpy()).T @ _bf16_round(Yt.cpu().numpy())
        assert np.abs(Wt.cpu().numpy() - ref).max() <= 1e-5 * np.sqrt(K) * 4 * np.abs(ref).max()


def test_wgrad_strips_without_partials_or_bias(H):
    """A strip problem (>= 512 tiles, twins) launched without sq_partials and without db: same dW as with both, nothing
    else written (the strip's partial / bias bookkeeping is optional like the one-tile kernel's)."""
    dev = "cuda"
    rng = np.random.RandomState(3)
    M, N, K = 2048, 1024, 128
    At = torch.tensor(rng.randn(K, M).astype(np.float32), device=dev)
    Yt = torch.tensor(rng.randn(K, N).astype(np.float32), device=dev)
    A16, Y16 = _bf16_twin(H, At), _bf16_twin(H, Yt)
    outs = []
    for full in (True, False):
        Wt = torch.full((M, N), float("nan"), device=dev)
        bt = torch.full((N,), float("nan"), device=dev)
        arr = (H.Wgrad * 1)(H.Wgrad(_p(At), _p(Yt), _p(Wt), _p(bt) if full else None, M, N, K, M, N, N, 0, 0, 0, 0, _p(A16), _p(Y16)))
        nblk = H.lib().air_wgrad_num_blocks(arr, 1)
        assert H.lib().air_wgrad_num_workgroups(arr, 1, 1) == nblk // 2
        part = torch.full((nblk,), float("nan"), device=dev)
        ist = torch.zeros(8, dtype=torch.int32, device=dev)
        H.check(H.lib().air_wgrad_grouped(arr, 1, 1, _p(part) if full else None, _p(ist) if full else None, _stream()))
        torch.cuda.synchronize()
        outs.append((Wt, bt, part, ist))
    assert torch.equal(outs[0][0], outs[1][0])
    assert bool(torch.isnan(outs[1][1]).all()) and bool(torch.isnan(outs[1][2]).all()) and int(outs[1][3][H.IST_GLOBAL_STEP]) == 0
    ref = Yt.double().sum(0)
    assert float((outs[0][1].double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) and int(outs[0][3][H.IST_GLOBAL_STEP]) == 1
    assert abs(float(outs[0][2].double().sum()) - float((outs[0][0].double() ** 2).sum() + (outs[0][1].double() ** 2).sum())) <= 1e-5 * float((outs[0][0].double() ** 2).sum())


_LDS_ORDER_SCRIPT = r'''
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tf-attend-infer-repeat_amd"))
from air import _hip as H
from air.transformer import transformer_grad
rng = np.random.RandomState(5)
B, N, Cc, w = 3, 2, int(sys.argv[3]), 28
att = np.zeros((N, B, H.ATT_STRIDE), np.float32)
att[..., H.ATT_S] = rng.uniform(0.15, 0.5, (N, B)); att[..., H.ATT_X] = rng.uniform(-0.9, 0.9, (N, B))
att[..., H.ATT_Y] = rng.uniform(-0.9, 0.9, (N, B)); att[..., H.ATT_Z] = rng.uniform(0.1, 1, (N, B)); att[..., H.ATT_MASK] = 1
g = (rng.randn(B, Cc * Cc) * np.where(rng.uniform(size=(B, Cc * Cc)) < 0.08, 1e7, 1e-2)).astype(np.float32)
vrec = rng.uniform(0.01, 0.99, (N, B, w * w)).astype(np.float32)
t = lambda a: torch.tensor(a, device="cuda")
att_d, g_d, v_d = t(att), t(g), t(vrec)
dgen, dsx = torch.zeros(N, B, w * w, device="cuda"), torch.zeros(N, B, 4, device="cuda")
p = lambda x: C.c_void_p(x.data_ptr())
wb = H.WriteBwd(p(g_d), p(v_d), p(att_d), p(dgen), p(dsx), B, N, Cc, w, 2, None, None, None, None)
H.check(H.lib().air_write_bwd(C.byref(wb), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
U = t(rng.uniform(0, 1, (B, 28, 28)).astype(np.float32))
th = t((np.tile(np.array([[1.4, 0.2, 0.5], [-0.1, 1.5, -0.6]], np.float32), (B, 1, 1)) + rng.randn(B, 2, 3).astype(np.float32) * 0.1))
d = t((rng.randn(B, 50, 50) * np.where(rng.uniform(size=(B, 50, 50)) < 0.1, 1e6, 1.0)).astype(np.float32))
dU, dth = transformer_grad(U, th, (50, 50), d)
torch.cuda.synchronize()
np.savez(sys.argv[2], dgen=dgen.cpu().numpy(), dsx=dsx.cpu().numpy(), dU=dU.cpu().numpy(), dth=dth.cpu().numpy())
'''


@pytest.mark.parametrize("Cc", [50, 128])
def test_lds_lane_order_probe_and_register_chain_fallback(H, tmp_path, Cc):
    """The graph-order backward relies on ds_add_f32 applying lanes in ascending order (undocumented on gfx950).
    The library probes that once per process, together with its first fallback, the DPP lane ring (one
    v_add_f32 ... wave_ror:1 per term; relies on the hardware interlocking a DPP read of the previous result).
    AIR_LDS_ORDER=0 forces "pipe not ordered" (air_write_bwd(literal=2) takes the rings), AIR_WB_RING=0 on top of it
    "ring not exact" (register chains, the last resort; air_transformer_bwd goes straight there): every path must give
    BIT-IDENTICAL results to the LDS-pipe path, so a part that behaves differently stays correct instead of changing
    gradients."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "lds_order.py"
    script.write_text(_LDS_ORDER_SCRIPT)
    outs = {}
    for mode in ("probe", "ring", "chains"):
        env = dict(os.environ)
        env.pop("AIR_LDS_ORDER", None)
        env.pop("AIR_WB_RING", None)
        if mode != "probe":
            env["AIR_LDS_ORDER"] = "0"
        if mode == "chains":
            env["AIR_WB_RING"] = "0"
        out = tmp_path / ("out_%s.npz" % mode)
        r = subprocess.run([sys.executable, str(script), root, str(out), str(Cc)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "lane order differs" not in r.stderr          # the probe finds MI355X ordered
        assert "lane ring is not" not in r.stderr            # ... and its ring exact
        outs[mode] = np.load(out)
    for mode in ("ring", "chains"):
        for k in ("dgen", "dsx", "dU", "dth"):
            assert np.array_equal(outs["probe"][k], outs[mode][k]), (mode, k)
    assert np.abs(outs["probe"]["dgen"]).max() > 1.0            # the residue was there


def test_gemm_throughput_tiling_for_the_deep_input_product(H):
    """tile (8, 4): the throughput kernel for the hoisted x.Wx of a large canvas (fp32 image batch x bf16
    shadow of Wx, split-K slabs; air_model.py:286 recomputed once per step).  Sum of its slabs vs the bf16-rounded fp64
    product; ineligible shapes are refused (the caller falls back to the latency tiles)."""
    rng = np.random.RandomState(31)
    for M, N, K, ks in ((256, 1024, 16384, 8), (128, 128, 1024, 4), (256, 64, 4096, 8)):
        A = torch.tensor(rng.uniform(0, 1, (M, K)).astype(np.float32), device="cuda")
        B = torch.tensor(rng.uniform(-0.05, 0.05, (K, N)).astype(np.float32), device="cuda")
        B16 = _bf16_twin(H, B)
        S = H.lib().air_gemm_slabs(K, ks)
        Ct = torch.full((S, M, N), float("nan"), device="cuda")
        g = _gemm_struct(H, A, B, Ct, M, N, K, K, N, N, 1, ksplit=ks, tile_m=8, tile_n=4, B16=B16)
        # 64 x 128 tiles where they still give every CU a workgroup (the 128 x 128 step's shape), 64 x 64 otherwise
        assert _kernel_name(H, g) == ("gemm_xw_tp_kernel<128>" if (N // 128) * (M // 64) * S >= 256 and N % 128 == 0 else "gemm_xw_tp_kernel<64>")
        H.check(H.lib().air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        got = Ct.double().sum(0).cpu().numpy()
        ref = _ref_gemm(A.cpu().numpy(), B.cpu().numpy(), 0, 0, 1)
        assert np.abs(got - ref).max() / np.sqrt(K) < 2e-5, (M, N, K)
        # and it agrees with the latency tiles on the same twins up to the fp32 summation order
        C2 = torch.zeros((S, M, N), device="cuda")
        g2 = _gemm_struct(H, A, B, C2, M, N, K, K, N, N, 1, ksplit=ks, tile_m=4, tile_n=4, B16=B16)
        H.check(H.lib().air_gemm(C.byref(g2), _stream()))
        torch.cuda.synchronize()
        assert np.abs(got - C2.double().sum(0).cpu().numpy()).max() < 1e-3 * np.abs(ref).max()
    # refused: ragged M, no shadow, no split
    A = torch.zeros(100, 1024, device="cuda"); B = torch.zeros(1024, 64, device="cuda"); Ct = torch.zeros(4, 100, 64, device="cuda")
    g = _gemm_struct(H, A, B, Ct, 100, 64, 1024, 1024, 64, 64, 1, ksplit=4, tile_m=8, tile_n=4, B16=_bf16_twin(H, B))
    assert H.lib().air_gemm(C.byref(g), _stream()) == -3
    A = torch.zeros(128, 1024, device="cuda"); Ct = torch.zeros(4, 128, 64, device="cuda")
    g = _gemm_struct(H, A, B, Ct, 128, 64, 1024, 1024, 64, 64, 1, ksplit=4, tile_m=8, tile_n=4)
    assert H.lib().air_gemm(C.byref(g), _stream()) == -1


# ---- panel-blocked bf16 twins of the weights (air_panel_t): layout, who maintains them, and the GEMMs that read them ----

def _panel_ref(W, gates):
    """numpy statement of air_panel_t's layout for one row-major [K, N] matrix (include/air_hip.h): bf16 bit patterns as int16"""
    K, N = W.shape
    w16 = torch.tensor(W).to(torch.bfloat16).view(torch.int16).numpy()
    if gates:
        R = N // 4
        out = np.zeros((R // 4, K, 16), np.int16)
        for gate in range(4):
            blk = w16[:, gate * R:(gate + 1) * R].reshape(K, R // 4, 4)             # [k][panel][unit % 4]
            out[:, :, gate * 4:gate * 4 + 4] = blk.transpose(1, 0, 2)
        return out.reshape(-1)
    P = (N + 15) // 16
    pad = np.zeros((K, P * 16), np.int16)
    pad[:, :N] = w16
    return pad.reshape(K, P, 16).transpose(1, 0, 2).reshape(-1)


def _panel_setup(H, rng, mats):
    """flat fp32 buffer holding `mats` = [(K, N, gates, exclusive)] with gaps (biases) between them, + the descriptors"""
    off, doff, items, pans = 8, 0, [], []
    for K, N, gates, excl in mats:
        items.append((off, doff))
        pans.append(H.Panel(off, doff, K, N, 4 if gates else 0, 1 if excl else 0))
        off += (K * N + 7) // 8 * 8 + 24                  # a bias-sized gap that belongs to no matrix
        doff += ((K * N if gates else (N + 15) // 16 * 16 * K) + 7) // 8 * 8
    flat = rng.uniform(-1, 1, off).astype(np.float32)
    return flat, items, (H.Panel * len(pans))(*pans), doff


PANEL_MATS = [(2500, 1024, True, True), (256, 1024, True, False), (256, 320, False, False), (784, 512, False, False),
              (50, 104, False, False), (33, 8, False, False)]


def test_panel_shadow_layout_and_adam_maintains_it(H):
    """air_panel_shadow writes bf16(params) in the documented panel layout (plain 16-column panels, the last one
    zero-padded; gate-interleaved panels for an LSTM kernel, air_model.py:286 i, j, f, o blocks), and
    air_adam_clip_step_panels keeps BOTH twins current with the variables it updates: the panel twin everywhere, the
    row-major twin everywhere except over `exclusive` matrices (left untouched).  ApplyAdam itself is unchanged:
    variables / slots bit-identical to air_adam_clip_step."""
    rng = np.random.RandomState(41)
    flat, items, pans, ptotal = _panel_setup(H, rng, PANEL_MATS)
    n = flat.size
    p = torch.tensor(flat, device="cuda")
    pan = torch.full((ptotal,), 0x7fff, dtype=torch.int16, device="cuda")
    pan.zero_()
    H.check(H.lib().air_panel_shadow(_p(p), _p(pan), pans, len(pans), _stream()))
    torch.cuda.synchronize()
    got = pan.cpu().numpy()
    for (K, N, gates, excl), (off, doff) in zip(PANEL_MATS, items):
        ref = _panel_ref(flat[off:off + K * N].reshape(K, N), gates)
        assert np.array_equal(got[doff:doff + ref.size], ref), (K, N, gates)
    # one Adam step with real gradients, twice: with and without the panel tables
    g = torch.tensor(rng.uniform(-1, 1, n).astype(np.float32), device="cuda")
    dyn = np.zeros(H.DYN_COUNT, np.float32)
    dyn[H.DYN_LEARNING_RATE], dyn[H.DYN_CLIP_NORM] = 1e-2, 1.0
    dyn_d = torch.tensor(dyn, device="cuda")
    res = {}
    for mode in ("plain", "panels"):
        pp, m, v = p.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        ist = torch.zeros(H.IST_COUNT, dtype=torch.int32, device="cuda")
        npart = H.lib().air_optim_num_partials(n)
        part = torch.zeros(npart, device="cuda")
        gn = torch.zeros(1, device="cuda")
        sh = torch.full((n,), 0x1234, dtype=torch.int16, device="cuda")
        pn = torch.full((ptotal,), 0x1234, dtype=torch.int16, device="cuda")
        H.check(H.lib().air_grad_sqnorm(_p(g), n, _p(part), _p(ist), _stream()))
        if mode == "plain":
            H.check(H.lib().air_adam_clip_step(_p(pp), _p(g), _p(m), _p(v), n, _p(part), npart, _p(dyn_d), _p(ist), 1.0, 0.9, 0.999, 1e-8,
                                               _p(sh), _p(gn), _stream()))
        else:
            H.check(H.lib().air_adam_clip_step_panels(_p(pp), _p(g), _p(m), _p(v), n, _p(part), npart, _p(dyn_d), _p(ist), 1.0, 0.9, 0.999,
                                                      1e-8, _p(sh), pans, len(pans), _p(pn), _p(gn), _stream()))
        torch.cuda.synchronize()
        res[mode] = (pp, m, v, sh, pn, float(gn))
    for a, b in zip(res["plain"][:3], res["panels"][:3]):
        assert torch.equal(a, b)
    assert res["plain"][5] == res["panels"][5]
    pp, sh, pn = res["panels"][0].cpu().numpy(), res["panels"][3].cpu().numpy(), res["panels"][4].cpu().numpy()
    assert not np.array_equal(pp, flat)
    flat16 = torch.tensor(pp).to(torch.bfloat16).view(torch.int16).numpy()
    assert np.array_equal(res["plain"][3].cpu().numpy(), flat16)                    # the plain call: the whole row-major twin
    keep = np.ones(n, bool)
    for (K, N, gates, excl), (off, doff) in zip(PANEL_MATS, items):
        ref = _panel_ref(pp[off:off + K * N].reshape(K, N), gates)
        sl = pn[doff:doff + ref.size]
        if gates or N % 16 == 0:
            assert np.array_equal(sl, ref), (K, N, gates)
        else:                                             # the pad columns of the last panel are never written
            P = (N + 15) // 16
            m_ = np.zeros((K, P * 16), bool); m_[:, :N] = True
            m_ = m_.reshape(K, P, 16).transpose(1, 0, 2).reshape(-1)
            assert np.array_equal(sl[m_], ref[m_]) and np.all(sl[~m_] == 0x1234)
        if excl:
            keep[off:off + K * N] = False
            assert np.all(sh[off:off + K * N] == 0x1234)                             # exclusive: the row-major twin is left alone
    assert np.array_equal(sh[keep], flat16[keep])
    # argument errors: overlapping / unordered matrices, unaligned offsets, too many
    bad = (H.Panel * 2)(H.Panel(0, 0, 4, 8, 0, 0), H.Panel(16, 64, 4, 8, 0, 0))
    assert H.lib().air_panel_shadow(_p(p), _p(pan), bad, 2, _stream()) == -1
    bad = (H.Panel * 1)(H.Panel(2, 0, 4, 8, 0, 0))
    assert H.lib().air_panel_shadow(_p(p), _p(pan), bad, 1, _stream()) == -3
    bad = (H.Panel * 1)(H.Panel(0, 0, 4, 24, 4, 0))                                   # gates: N % 16
    assert H.lib().air_panel_shadow(_p(p), _p(pan), bad, 1, _stream()) == -1
    assert H.lib().air_panel_shadow(_p(p), _p(pan), pans, 17, _stream()) == -2


def _panels_of(H, W, gates):
    """device panel twin of one device matrix through the ABI"""
    K, N = W.shape
    size = K * N if gates else (N + 15) // 16 * 16 * K
    out = torch.zeros(size, dtype=torch.int16, device="cuda")
    pd = (H.Panel * 1)(H.Panel(0, 0, K, N, 4 if gates else 0, 0))
    H.check(H.lib().air_panel_shadow(_p(W), _p(out), pd, 1, _stream()))
    return out


@pytest.mark.parametrize("M,N,K,tile", [(192, 320, 256, (0, 0)), (192, 512, 784, (0, 0)), (192, 256, 512, (0, 0)), (192, 784, 512, (0, 0)),
                                        (1280, 512, 784, (0, 0)), (50, 104, 2048, (1, 1)), (37, 200, 1160, (2, 2)),
                                        (64, 1024, 2496, (2, 2))])
def test_gemm_reads_panel_blocked_weights_bit_identically(H, M, N, K, tile):
    """air_gemm_t.B16p: the same product from the panel-blocked twin of B (16- and 32-column tiles) as from the row-major
    twin -- same bf16 values, same k order, same reduction: BIT-IDENTICAL, with or without the row-major twin present."""
    rng = np.random.RandomState(M + N + K)
    A = torch.tensor(rng.uniform(-1, 1, (M, K)).astype(np.float32), device="cuda")
    B = torch.tensor(rng.uniform(-1, 1, (K, N)).astype(np.float32), device="cuda")
    bias = torch.tensor(rng.uniform(-1, 1, N).astype(np.float32), device="cuda")
    A16, B16, B16p = _bf16_twin(H, A), _bf16_twin(H, B), _panels_of(H, B, False)
    outs = []
    for kw in (dict(B16=B16), dict(B16p=B16p), dict(B16=B16, B16p=B16p)):
        Ct = torch.full((M, N), float("nan"), device="cuda")
        g = _gemm_struct(H, A, B, Ct, M, N, K, K, N, N, 1, tile_m=tile[0], tile_n=tile[1], bias=bias, act=H.ACT_SOFTPLUS, A16=A16, **kw)
        assert _kernel_name(H, g).startswith("gemm_bf16tw_kernel")
        H.check(H.lib().air_gemm(C.byref(g), _stream()))
        torch.cuda.synchronize()
        outs.append(Ct)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert bool(torch.isfinite(outs[0]).all())


def test_gemm_lstm_tiles_read_gate_interleaved_panels_bit_identically(H):
    """The four-unit x four-gate tiles (AIR_EPI_LSTM_FWD; AIR_EPI_LSTM_FWD0 = the hoisted x.Wx carrying the first step)
    on the gate-interleaved panel twin of the LSTM kernel: every output bit-identical to the row-major twin and to the
    fp32-operand kernels; x.Wx as fp32 A (the caller's image batch) and as a bf16 twin; one x.Wx slab as the addend of
    the later steps."""
    dev, lib = "cuda", H.lib()
    rng = np.random.RandomState(43)
    f = lambda *s: torch.tensor(rng.uniform(-1, 1, s).astype(np.float32), device=dev)  # noqa: E731
    i16 = lambda *s: torch.zeros(*s, dtype=torch.int16, device=dev)  # noqa: E731
    for Bn, R, D in ((64, 256, 2500), (48, 64, 520)):
        X, Wx, Wh, bias = f(Bn, D).abs(), f(D, 4 * R) * 0.05, f(R, 4 * R) * 0.1, f(4 * R) * 0.1
        Wx16, WxP, Wh16, WhP, X16 = _bf16_twin(H, Wx), _panels_of(H, Wx, True), _bf16_twin(H, Wh), _panels_of(H, Wh, True), _bf16_twin(H, X)
        res = []
        for kw in (dict(), dict(B16=Wx16), dict(B16p=WxP), dict(B16p=WxP, A16=X16)):
            xw, acts, c1, h1, h16 = f(Bn, 4 * R), f(Bn, 4 * R), f(Bn, R), f(Bn, R), i16(Bn, R)
            g = _gemm_struct(H, X, Wx, xw, Bn, 4 * R, D, D, 4 * R, 4 * R, 1, bias=bias, epi=H.EPI_LSTM_FWD0, q0=acts, q1=c1, q2=h1,
                             q2_16=h16, **kw)
            name = _kernel_name(H, g)
            if not kw:
                assert name.startswith("gemm_bf16v2_kernel"), name
            elif D % 8 or "A16" not in kw:
                assert name.startswith("gemm_bf16tw_kernel<1, 1, false, 6, true, ") or (D % 8 and "A16" in kw), name
            H.check(lib.air_gemm(C.byref(g), _stream()))
            torch.cuda.synchronize()
            assert torch.equal(h16.view(torch.bfloat16), h1.to(torch.bfloat16))
            res.append((xw, acts, c1, h1))
        for other in res[1:]:
            for a0, a1 in zip(res[0], other):
                assert torch.equal(a0, a1)
        xw0, _, c_prev, h_prev = res[0]
        # a later step: h.Wh + the ONE slab of x.Wx + bias -> gates
        h16_in = _bf16_twin(H, h_prev)
        res2 = []
        for kw in (dict(B16=Wh16), dict(B16p=WhP), dict(B16=Wh16, B16p=WhP)):
            acts, c1, h1, dummy, h16 = f(Bn, 4 * R), f(Bn, R), f(Bn, R), f(Bn, 4 * R), i16(Bn, R)
            g = _gemm_struct(H, h_prev, Wh, dummy, Bn, 4 * R, R, R, 4 * R, 4 * R, 1, bias=bias, addend=xw0, ldadd=4 * R, addend_slabs=1,
                             epi=H.EPI_LSTM_FWD, p0=c_prev, q0=acts, q1=c1, q2=h1, q2_16=h16, A16=h16_in, **kw)
            assert "tw_kernel" in _kernel_name(H, g)
            H.check(lib.air_gemm(C.byref(g), _stream()))
            torch.cuda.synchronize()
            res2.append((acts, c1, h1))
        for other in res2[1:]:
            for a0, a1 in zip(res2[0], other):
                assert torch.equal(a0, a1)
        # ... against the float64 statement of BasicLSTMCell on the bf16-rounded operands
        pre = _bf16_round(h_prev.cpu().numpy()) @ _bf16_round(Wh.cpu().numpy()) + xw0.cpu().numpy().astype(np.float64) + bias.cpu().numpy()
        sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
        cn = c_prev.cpu().numpy() * sig(pre[:, 2 * R:3 * R] + 1.0) + sig(pre[:, :R]) * np.tanh(pre[:, R:2 * R])
        np.testing.assert_allclose(res2[0][1].cpu().numpy(), cn, atol=5e-5)


@pytest.mark.parametrize("Cc,w,N,B", [(128, 28, 5, 64), (50, 28, 3, 16)])
def test_write_bwd_takes_its_items_longest_first_without_changing_results(H, Cc, w, N, B):
    """air_write_fwd_t.wb_order: one extra workgroup of the compose launch sorts the (image, step) items by the corner
    terms the graph-order write backward will accumulate for them (inactive items last) -- a permutation of 0 .. N*B-1,
    non-increasing in the cost class (1024 terms); air_write_bwd_t.order: the backward computes item order[i] in workgroup i.  Gradients wrt
    the decoder output and wrt (s, x, y, z) and the batch means are BIT-IDENTICAL with and without the order
    (transformer.py:56-117 under tf.gradients; air_model.py:595-611)."""
    rng = np.random.RandomState(Cc + B)
    lib = H.lib()
    Z, NB = 50, N * B
    att = np.zeros((N, B, H.ATT_STRIDE), np.float32)
    att[:, :, H.ATT_S] = rng.uniform(0.08, 0.9, (N, B)); att[:, :, H.ATT_X] = rng.uniform(-0.9, 0.9, (N, B))
    att[:, :, H.ATT_Y] = rng.uniform(-0.9, 0.9, (N, B)); att[:, :, H.ATT_Z] = rng.uniform(0.2, 1.0, (N, B))
    alive = np.cumprod(rng.uniform(0, 1, (N, B)) < 0.75, axis=0).astype(np.float32)
    att[:, :, H.ATT_MASK] = alive
    att[1:, :, H.ATT_MASK_PREV] = alive[:-1]; att[0, :, H.ATT_MASK_PREV] = 1.0
    vrec = rng.uniform(0.05, 0.95, (N, B, w * w)).astype(np.float32)
    ml = rng.uniform(-1, 1, (N, B, 2 * Z)).astype(np.float32)
    images = (rng.uniform(0, 1, (B, Cc * Cc)) * (rng.uniform(0, 1, (B, Cc * Cc)) < 0.2)).astype(np.float32)
    targets = rng.randint(0, N + 1, B).astype(np.int32)
    dyn = np.zeros(H.DYN_COUNT, np.float32)
    dyn[H.DYN_VAE_PV], dyn[H.DYN_GRAD_SCALE] = 1.0, 1.0 / B
    t = lambda a_, dt=torch.float32: torch.tensor(a_, dtype=dt, device="cuda")  # noqa: E731
    vrec_d, ml_d, img_d, dyn_d, tg_d = t(vrec), t(ml), t(images), t(dyn), t(targets, torch.int32)
    res = {}
    for ordered in (False, True):
        att_d = t(att)
        recon, d_recon = torch.zeros(B, Cc * Cc, device="cuda"), torch.zeros(B, Cc * Cc, device="cuda")
        rec_loss, run_loss, loss_item, scal = (torch.full((n_,), 7.0, device="cuda") for n_ in (B, B, B, 4))
        digits = torch.zeros(B, dtype=torch.int32, device="cuda")
        order = torch.full((NB,), -1, dtype=torch.int32, device="cuda") if ordered else None
        wf = H.WriteFwd(_p(vrec_d), _p(ml_d), _p(img_d), _p(dyn_d), _p(att_d), _p(recon), _p(rec_loss), _p(d_recon),
                        _p(run_loss), _p(digits), _p(loss_item), B, N, Cc, w, Z, _p(order))
        H.check(lib.air_write_fwd(C.byref(wf), _stream()), "air_write_fwd")
        dgen, dsx = torch.full((N, B, w * w), 7.0, device="cuda"), torch.full((N, B, 4), 7.0, device="cuda")
        wb = H.WriteBwd(_p(d_recon), _p(vrec_d), _p(att_d), _p(dgen), _p(dsx), B, N, Cc, w, 2, _p(loss_item), _p(tg_d), _p(digits), _p(scal),
                        None, _p(order))
        H.check(lib.air_write_bwd(C.byref(wb), _stream()), "air_write_bwd")
        torch.cuda.synchronize()
        res[ordered] = (recon, d_recon, rec_loss, loss_item, dgen, dsx, scal[:2].clone())
        if ordered:
            o = order.cpu().numpy()
            assert sorted(o.tolist()) == list(range(NB))                                   # a permutation
            act = alive.reshape(-1)[o] != 0
            n_act = int(act.sum())
            assert act[:n_act].all() and not act[n_act:].any()                             # inactive items last
            # the cost the sort used, restated: out-of-range columns x rows of the write transformer (transformer.py:75-87)
            def oob(sv, shift):
                tt = np.linspace(-1, 1, Cc, dtype=np.float32)
                X = ((np.float32(1.0) / sv * tt + (-shift) / sv) + np.float32(1.0)) * np.float32(w - 1.001) / np.float32(2.0)
                f0 = np.floor(X)
                return int((np.clip(f0, 0, w - 1) == np.clip(f0 + 1, 0, w - 1)).sum())
            cost = []
            for it in o[:n_act]:
                a_ = att.reshape(NB, -1)[it]
                cost.append(oob(a_[H.ATT_S], a_[H.ATT_X]) * oob(a_[H.ATT_S], a_[H.ATT_Y]))
            # ... in classes of 1024 terms (a counting sort: within a class the order is free)
            klass = [(4 * c_ + 12000) >> 10 for c_ in cost]
            assert all(c0 >= c1 for c0, c1 in zip(klass, klass[1:])), "items are not longest first"
            assert klass[0] > klass[-1]
    for a0, a1 in zip(res[False], res[True]):
        assert torch.equal(a0, a1)
    assert float(res[True][4].abs().max()) > 0 and not bool((res[True][5] == 7.0).any())
    # refused: an order with a backward that is not the graph-order one; too many items to sort
    wb = H.WriteBwd(_p(d_recon), _p(vrec_d), _p(att_d), _p(dgen), _p(dsx), B, N, Cc, w, 0, None, None, None, None, None, _p(order))
    assert lib.air_write_bwd(C.byref(wb), _stream()) == -1


@pytest.mark.parametrize("M,Hd,Z", [(192, 256, 50), (37, 192, 50), (16, 256, 52), (5, 64, 2)])
def test_vae_bottleneck_forward_exact_fp32(H, M, Hd, Z):
    """air_bottleneck_fwd_t.exact_fp32: vae.py:16-30 in one launch with exact fp32 products (the fp32 path's arithmetic):
    against float64 on the fp32 operands, and against the two exact-fp32 air_gemm launches it replaces."""
    rng = np.random.RandomState(5)
    K1 = 256
    X = rng.uniform(0, 2, (M, K1)).astype(np.float32)
    Wml = rng.uniform(-0.1, 0.1, (K1, 2 * Z)).astype(np.float32)
    bml = rng.uniform(-0.1, 0.1, 2 * Z).astype(np.float32)
    eps = rng.randn(M, Z).astype(np.float32)
    Wg = rng.uniform(-0.3, 0.3, (Z, Hd)).astype(np.float32)
    bg = rng.uniform(-0.1, 0.1, Hd).astype(np.float32)
    t = {k: torch.tensor(v, device="cuda") for k, v in dict(X=X, Wml=Wml, bml=bml, eps=eps, Wg=Wg, bg=bg).items()}
    ml = torch.full((M, 2 * Z), float("nan"), device="cuda")
    z = torch.full((M, Z), float("nan"), device="cuda")
    g = torch.full((M, Hd), float("nan"), device="cuda")
    a = H.BottleneckFwd(_p(t["X"]), _p(t["Wml"]), _p(t["bml"]), _p(t["eps"]), _p(t["Wg"]), _p(t["bg"]), _p(ml), _p(z), _p(g),
                        M, K1, Z, Hd, K1, None, None, None, None, None, 1)
    H.check(H.lib().air_vae_bottleneck_fwd(C.byref(a), _stream()))
    torch.cuda.synchronize()
    ml_ref = X.astype(np.float64) @ Wml.astype(np.float64) + bml
    z_ref = ml_ref[:, :Z] + eps * np.sqrt(np.exp(ml_ref[:, Z:]))
    assert np.abs(ml.cpu().numpy() - ml_ref).max() < 3e-6
    # (the sample amplifies the log-variance's last bits by eps * exp(lv / 2) / 2: relative to its own magnitude)
    assert np.all(np.abs(z.cpu().numpy() - z_ref) <= 4e-6 * np.maximum(1.0, np.abs(z_ref)))
    g_ref = _softplus_tf(z.cpu().numpy().astype(np.float64) @ Wg.astype(np.float64) + bg)
    assert np.all(np.abs(g.cpu().numpy() - g_ref) <= 4e-6 * np.maximum(1.0, np.abs(g_ref)))
    ml2, z2, g2 = torch.empty_like(ml), torch.empty_like(z), torch.empty_like(g)
    g1 = _gemm_struct(H, t["X"], t["Wml"], ml2, M, 2 * Z, K1, K1, 2 * Z, 2 * Z, 0, bias=t["bml"], epi=H.EPI_REPARAM_FWD,
                      p0=t["eps"], q0=z2)
    H.check(H.lib().air_gemm(C.byref(g1), _stream()))
    g2s = _gemm_struct(H, z2, t["Wg"], g2, M, Hd, Z, Z, Hd, Hd, 0, bias=t["bg"], act=H.ACT_SOFTPLUS)
    H.check(H.lib().air_gemm(C.byref(g2s), _stream()))
    torch.cuda.synchronize()
    # (two fp32 summation orders of a K = 256 contraction: each within 3e-6 of float64)
    assert (ml - ml2).abs().max() < 6e-6
    assert bool(((z - z2).abs() <= 1e-5 * z2.abs().clamp(min=1.0)).all()) and bool(((g - g2).abs() <= 2e-5 * g2.abs().clamp(min=1.0)).all())
    # limit of the exact form: Wml rows of at most 104 floats in LDS
    a.Z = 54
    assert H.lib().air_vae_bottleneck_fwd(C.byref(a), _stream()) == -2


@pytest.mark.parametrize("M,K1,Z", [(192, 256, 50), (37, 200, 50), (16, 64, 64), (5, 256, 2)])
def test_vae_bottleneck_backward_exact_fp32(H, M, K1, Z):
    """air_bottleneck_bwd_t.exact_fp32: d_z = dG.Wg^T, the reparameterisation + KL gradients, d_x = (d_ml.Wml^T) *
    softplus'(x) with exact fp32 products: against float64 and against the two exact-fp32 launches it replaces."""
    rng = np.random.RandomState(6)
    Hd = 256
    dG = rng.randn(M, Hd).astype(np.float32) * 0.1
    Wg = rng.uniform(-0.3, 0.3, (Z, Hd)).astype(np.float32)
    ml = rng.uniform(-1, 1, (M, 2 * Z)).astype(np.float32)
    eps = rng.randn(M, Z).astype(np.float32)
    att = np.zeros((M, H.ATT_STRIDE), np.float32)
    att[:, H.ATT_MASK] = rng.randint(0, 2, M)
    dyn = np.zeros(32, np.float32)
    dyn[H.DYN_GRAD_SCALE], dyn[H.DYN_VAE_PV], dyn[H.DYN_VAE_PM] = 1.0 / 64, 0.8, 0.1
    Wml = rng.uniform(-0.1, 0.1, (K1, 2 * Z)).astype(np.float32)
    x = rng.uniform(0.01, 2, (M, K1)).astype(np.float32)
    t = {k: torch.tensor(v, device="cuda") for k, v in dict(dG=dG, Wg=Wg, ml=ml, eps=eps, att=att, dyn=dyn, Wml=Wml, x=x).items()}
    d_ml = torch.full((M, 2 * Z), float("nan"), device="cuda")
    d_x = torch.full((M, K1), float("nan"), device="cuda")
    a = H.BottleneckBwd(_p(t["dG"]), _p(t["Wg"]), _p(t["ml"]), _p(t["eps"]), _p(t["att"]), _p(t["dyn"]), _p(t["Wml"]), _p(t["x"]),
                        _p(d_ml), _p(d_x), M, K1, Z, Hd, None, None, None, None, None, 1)
    H.check(H.lib().air_vae_bottleneck_bwd(C.byref(a), _stream()))
    torch.cuda.synchronize()
    dz = dG.astype(np.float64) @ Wg.astype(np.float64).T
    klg = att[:, H.ATT_MASK:H.ATT_MASK + 1].astype(np.float64) * dyn[H.DYN_GRAD_SCALE]
    var = np.exp(ml[:, Z:].astype(np.float64))
    dmean = dz + klg * (ml[:, :Z] - dyn[H.DYN_VAE_PM]) / dyn[H.DYN_VAE_PV]
    dlv = dz * eps * 0.5 * np.sqrt(var) + klg * 0.5 * (var / dyn[H.DYN_VAE_PV] - 1.0)
    ref = np.concatenate([dmean, dlv], 1)
    got = d_ml.cpu().numpy()
    assert np.abs(got - ref).max() < 3e-6
    dx_ref = (got.astype(np.float64) @ Wml.astype(np.float64).T) * (1.0 - np.exp(-x.astype(np.float64)))
    assert np.abs(d_x.cpu().numpy() - dx_ref).max() < 3e-6
    d_ml2, d_x2 = torch.empty_like(d_ml), torch.empty_like(d_x)
    g1 = _gemm_struct(H, t["dG"], t["Wg"], d_ml2, M, Z, Hd, Hd, Hd, 2 * Z, 0, transB=1, epi=H.EPI_REPARAM_BWD,
                      p0=t["ml"], p1=t["eps"], p2=t["att"], p3=t["dyn"])
    H.check(H.lib().air_gemm(C.byref(g1), _stream()))
    g2 = _gemm_struct(H, d_ml2, t["Wml"], d_x2, M, K1, 2 * Z, 2 * Z, 2 * Z, K1, 0, transB=1, aux=t["x"], ldaux=K1,
                      actgrad=H.GRAD_SOFTPLUS)
    H.check(H.lib().air_gemm(C.byref(g2), _stream()))
    torch.cuda.synchronize()
    assert (d_ml - d_ml2).abs().max() < 6e-6 and (d_x - d_x2).abs().max() < 6e-6


def test_wgrad_twins_of_padded_rows_ignore_what_the_pad_holds(H):
    """air_wgrad_t with M % 4 != 0 and rows padded to a multiple of 4 (M = 50, lda = 52 -- z of the first generative layer,
    air_bottleneck_fwd_t.ldz): the 8-byte twin pieces of the last quad read the two pad columns.  The pad must be READABLE
    and may hold ANYTHING: with NaN / huge garbage in the pad columns of A and A16, dW, db and the global-norm partials
    are bit-identical to the fp32-operand path on unpadded rows (accumulator rows >= M are never stored nor squared)."""
    dev = "cuda"
    rng = np.random.RandomState(12)
    M, N, K, lda = 50, 256, 192, 52
    A = rng.randn(K, M).astype(np.float32)
    dY = (rng.randn(K, N) * 10.0 ** rng.randint(-2, 3, (K, 1))).astype(np.float32)
    A_d, dY_d = torch.tensor(A, device=dev), torch.tensor(dY, device=dev)
    outs = []
    for padded in (False, True):
        if padded:
            Ap = torch.full((K, lda), float("nan"), device=dev)
            Ap[:, M] = 3.0e38
            Ap[:, :M] = A_d
            A16 = _bf16_twin(H, Ap.contiguous())
            A16.view(K, lda)[:, M:] = torch.tensor([0x7FC0, 0x7F7F], dtype=torch.int16, device=dev)     # bf16 NaN, bf16 max
            src, ld, tw = Ap, lda, (A16, _bf16_twin(H, dY_d))
        else:
            src, ld, tw = A_d, M, (None, None)
        dW, db = torch.full((M, N), float("nan"), device=dev), torch.full((N,), float("nan"), device=dev)
        arr = (H.Wgrad * 1)(H.Wgrad(_p(src), _p(dY_d), _p(dW), _p(db), M, N, K, ld, N, N, 0, 0, 0, 0, _p(tw[0]), _p(tw[1])))
        nblk = H.lib().air_wgrad_num_blocks(arr, 1)
        part = torch.full((nblk,), float("nan"), device=dev)
        ist = torch.zeros(8, dtype=torch.int32, device=dev)
        H.check(H.lib().air_wgrad_grouped(arr, 1, 1, _p(part), _p(ist), _stream()), "air_wgrad_grouped")
        torch.cuda.synchronize()
        outs.append((dW, db, part))
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    ref = _bf16_round(A).T.astype(np.float64) @ _bf16_round(dY).astype(np.float64)
    assert np.abs(outs[1][0].cpu().numpy() - ref).max() <= 1e-5 * np.sqrt(K) * 4 * np.abs(ref).max()


@pytest.mark.parametrize("exact", [0, 1])
def test_vae_bottleneck_forward_padded_z_rows_leave_the_pad_alone(H, exact):
    """air_bottleneck_fwd_t.ldz = 52 for Z = 50: z (and its bf16 twin) are written with that row stride, the same values
    as with ldz = 0, and the two pad columns of every row keep what they held."""
    rng = np.random.RandomState(13)
    M, K1, Z, Hd, ldz = 37, 256, 50, 256, 52
    t = {k: torch.tensor(v.astype(np.float32), device="cuda") for k, v in dict(
        X=rng.uniform(0, 2, (M, K1)), Wml=rng.uniform(-0.1, 0.1, (K1, 2 * Z)), bml=rng.uniform(-0.1, 0.1, 2 * Z),
        eps=rng.randn(M, Z), Wg=rng.uniform(-0.3, 0.3, (Z, Hd)), bg=rng.uniform(-0.1, 0.1, Hd)).items()}
    res = []
    for ld in (0, ldz):
        w = ld or Z
        ml, g = torch.full((M, 2 * Z), float("nan"), device="cuda"), torch.full((M, Hd), float("nan"), device="cuda")
        z = torch.full((M, w), 7.0, device="cuda")
        z16 = torch.full((M, w), 0x1234, dtype=torch.int16, device="cuda")
        g16 = torch.zeros(M, Hd, dtype=torch.int16, device="cuda")
        a = H.BottleneckFwd(_p(t["X"]), _p(t["Wml"]), _p(t["bml"]), _p(t["eps"]), _p(t["Wg"]), _p(t["bg"]), _p(ml), _p(z), _p(g),
                            M, K1, Z, Hd, K1, None if exact else _p(z16), None if exact else _p(g16), None, None, None, exact, ld)
        H.check(H.lib().air_vae_bottleneck_fwd(C.byref(a), _stream()))
        torch.cuda.synchronize()
        res.append((ml, z, g, z16))
    (ml0, z0, g0, z160), (ml1, z1, g1, z161) = res
    assert torch.equal(ml0, ml1) and torch.equal(g0, g1) and torch.equal(z0, z1[:, :Z])
    assert bool((z1[:, Z:] == 7.0).all())
    if not exact:
        assert torch.equal(z160, z161[:, :Z]) and bool((z161[:, Z:] == 0x1234).all())
        assert torch.equal(z161[:, :Z].contiguous().view(torch.bfloat16), z1[:, :Z].to(torch.bfloat16))
