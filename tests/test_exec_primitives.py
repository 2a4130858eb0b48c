"""Independent check of oracle/graphdef_exec.py's op kernels (test infrastructure).

The oracle is pinned to the reference's saved graph as EXECUTED by graphdef_exec -- whose ~110 numpy
kernels are this build's own restatements of TensorFlow 1.3's (TF cannot run here).  A primitive that
both the executor and the oracle mis-stated the same way would go unnoticed, so every executor kernel that
touches floats is compared here with the corresponding torch-CPU op -- a third implementation, written
by neither -- on random and edge inputs (zeros, +-0, ties of Round, the Softplus threshold region, large
|x| for Sigmoid/Tanh, negative operands of FloorMod).  Tolerances: exact for selection / rounding /
comparison ops; a few ulp for transcendental ones (numpy and torch call different libm kernels)."""
import numpy as np
import pytest
import torch

from oracle import graphdef_exec as gx


class _N:
    """stand-in for a graph node: attribute lookup only"""

    def __init__(self, **attr):
        self.attr = attr

    def a(self, k, default=None):
        return self.attr.get(k, default)


N0 = _N()
DT = [np.float32, np.float64]


def _edge(dtype, rng, n=4096, scale=4.0):
    x = (rng.randn(n) * scale).astype(dtype)
    special = np.array([0.0, -0.0, 0.5, -0.5, 1.5, 2.5, -1.5, -2.5, 1.0, -1.0, 13.9, -13.9, 14.1, -14.1, 15.9424, -15.9424,
                        16.0, -16.0, 30.0, -30.0, 60.0, -60.0, 88.0, -88.0, 1e-8, -1e-8, 1e-30, 3.4e5, -3.4e5], dtype)
    return np.concatenate([special, x])


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def _close(a, b, dtype, ulps=4, atol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    eps = np.finfo(dtype).eps
    bad = ~(np.abs(a - b) <= ulps * eps * np.abs(b) + atol) & ~(np.isnan(a) & np.isnan(b)) & ~(a == b)
    assert not bad.any(), (a[bad][:5], b[bad][:5])


@pytest.mark.parametrize("dtype", DT)
def test_unary_float_kernels(dtype):
    rng = np.random.RandomState(0)
    x = _edge(dtype, rng)
    pos = np.abs(x) + dtype(1e-3)
    tx = _t(x)
    tiny = float(np.finfo(dtype).tiny)
    exact = {"Neg": (x, -tx), "Floor": (x, torch.floor(tx)), "Round": (x, torch.round(tx)),      # half to even
             "Square": (x, tx * tx), "Relu": (x, torch.relu(tx)), "Reciprocal": (pos, 1.0 / _t(pos))}
    for op, (arg, want) in exact.items():
        got = gx.OPS[op](N0, arg)
        assert np.array_equal(got, want.numpy()), op
    _close(gx.OPS["Sqrt"](N0, pos), torch.sqrt(_t(pos)).numpy(), dtype, ulps=1)      # torch's vectorised sqrt: <= 1 ulp
    small = x[np.abs(x) < 80]
    _close(gx.OPS["Exp"](N0, small), torch.exp(_t(small)).numpy(), dtype)
    _close(gx.OPS["Log"](N0, pos), torch.log(_t(pos)).numpy(), dtype, atol=4 * np.finfo(dtype).eps)
    _close(gx.OPS["Tanh"](N0, x), torch.tanh(tx).numpy(), dtype, atol=tiny)
    # Sigmoid: 1 / (1 + exp(-x)) (Eigen scalar_sigmoid_op) vs torch's; exp(-x) overflows to inf -> exactly 0
    _close(gx.OPS["Sigmoid"](N0, x), torch.sigmoid(tx).numpy(), dtype, atol=tiny * 4)
    # Softplus: TF's two-sided threshold (log(eps) + 2) vs torch's log1p form with its one-sided one
    sp = gx.OPS["Softplus"](N0, x)
    _close(sp, torch.nn.functional.softplus(tx, threshold=-float(np.log(np.finfo(dtype).eps)) - 2.0).numpy(), dtype,
           ulps=8, atol=np.finfo(dtype).eps * 2)
    thr = -(np.log(np.finfo(dtype).eps) + 2.0)                           # softplus_op.h: above -threshold the input passes through
    assert (sp >= 0).all() and np.array_equal(sp[x > thr], x[x > thr])


@pytest.mark.parametrize("dtype", DT)
def test_binary_float_kernels(dtype):
    rng = np.random.RandomState(1)
    a, b = _edge(dtype, rng), _edge(dtype, np.random.RandomState(2))[::-1].copy()
    b_nz = np.where(b == 0, dtype(0.75), b)
    ta, tb, tbn = _t(a), _t(b), _t(b_nz)
    for op, want in (("Add", ta + tb), ("Sub", ta - tb), ("Mul", ta * tb), ("RealDiv", ta / tbn),
                     ("Maximum", torch.maximum(ta, tb)), ("Minimum", torch.minimum(ta, tb))):
        got = gx.OPS[op](N0, a, b_nz if op == "RealDiv" else b)
        assert np.array_equal(got, want.numpy(), equal_nan=True), op
    # FloorMod / FloorDiv: the result takes the sign of the divisor (Python semantics), as TF's
    _close(gx.OPS["FloorMod"](N0, a, b_nz), torch.remainder(ta, tbn).numpy(), dtype, ulps=2)
    assert np.array_equal(gx.OPS["FloorDiv"](N0, a, b_nz), torch.floor_divide(ta, tbn).numpy())
    ints_a, ints_b = np.array([7, -7, 7, -7, 0, 5], np.int32), np.array([3, 3, -3, -3, 4, 5], np.int32)
    assert np.array_equal(gx.OPS["FloorMod"](N0, ints_a, ints_b), torch.remainder(_t(ints_a), _t(ints_b)).numpy())
    assert np.array_equal(gx.OPS["FloorDiv"](N0, ints_a, ints_b),
                          torch.div(_t(ints_a), _t(ints_b), rounding_mode="floor").numpy())
    base = np.abs(a[np.abs(a) < 30]) + dtype(0.1)
    ex = (b[:len(base)] / 8).astype(dtype)
    _close(gx.OPS["Pow"](N0, base, ex), torch.pow(_t(base), _t(ex)).numpy(), dtype, ulps=16)
    for op, want in (("Less", ta < tb), ("LessEqual", ta <= tb), ("Greater", ta > tb), ("GreaterEqual", ta >= tb),
                     ("Equal", ta == tb)):
        assert np.array_equal(gx.OPS[op](N0, a, b), want.numpy()), op


@pytest.mark.parametrize("dtype", DT)
def test_gradient_kernels_match_torch_backward_ops(dtype):
    rng = np.random.RandomState(3)
    x, dy = _edge(dtype, rng), _edge(dtype, np.random.RandomState(4))
    tx, tdy = _t(x), _t(dy)
    y_sig, y_tanh = torch.sigmoid(tx), torch.tanh(tx)
    _close(gx.OPS["SigmoidGrad"](N0, y_sig.numpy(), dy), torch.ops.aten.sigmoid_backward(tdy, y_sig).numpy(), dtype)
    # dy * (1 - y*y): 1 - y*y cancels near |y| = 1, so the two agree to eps ABSOLUTE in that factor
    tg_a, tg_b = gx.OPS["TanhGrad"](N0, y_tanh.numpy(), dy), torch.ops.aten.tanh_backward(tdy, y_tanh).numpy()
    assert (np.abs(tg_a - tg_b) <= 4 * np.finfo(dtype).eps * np.abs(dy)).all()
    assert np.array_equal(gx.OPS["ReluGrad"](N0, dy, x), torch.ops.aten.threshold_backward(tdy, tx, 0.0).numpy())
    # SoftplusGrad: dy / (exp(-x) + 1) == dy * sigmoid(x); torch's kernel passes dy through above its threshold
    sel = np.abs(x) < 80
    _close(gx.OPS["SoftplusGrad"](N0, dy[sel], x[sel]),
           torch.ops.aten.softplus_backward(_t(dy[sel]), _t(x[sel]), 1.0, 1e9).numpy(), dtype,
           atol=float(np.finfo(dtype).tiny) * 8)
    pos = np.abs(x) + dtype(1e-3)
    tp = _t(pos).requires_grad_(True)
    ys = torch.sqrt(tp)
    ys.backward(tdy)
    _close(gx.OPS["SqrtGrad"](N0, ys.detach().numpy(), dy), tp.grad.numpy(), dtype)
    g2 = _t(rng.randn(37, 11).astype(dtype))
    assert np.allclose(gx.OPS["BiasAddGrad"](N0, g2.numpy()), g2.sum(0).numpy(), rtol=1e3 * np.finfo(dtype).eps, atol=1e-5)
    a = rng.randn(9, 5).astype(dtype)
    _close(gx.OPS["L2Loss"](N0, a), (torch.sum(_t(a) ** 2) / 2).numpy().astype(dtype), dtype, ulps=64)


@pytest.mark.parametrize("dtype", DT)
def test_contraction_reduction_and_scatter_kernels(dtype):
    rng = np.random.RandomState(5)
    tol = dict(rtol=2e3 * np.finfo(dtype).eps, atol=2e2 * np.finfo(dtype).eps)
    A, B = rng.randn(13, 29).astype(dtype), rng.randn(29, 7).astype(dtype)
    for ta in (False, True):
        for tb in (False, True):
            a, b = (A.T.copy() if ta else A), (B.T.copy() if tb else B)
            got = gx.OPS["MatMul"](_N(transpose_a=ta, transpose_b=tb), a, b)
            want = torch.matmul(_t(a).t() if ta else _t(a), _t(b).t() if tb else _t(b)).numpy()
            assert got.shape == (13, 7) and np.allclose(got, want, **tol)
    X, Y = rng.randn(4, 3, 6).astype(dtype), rng.randn(4, 6, 5).astype(dtype)
    assert np.allclose(gx.OPS["BatchMatMul"](_N(adj_x=False, adj_y=False), X, Y), torch.bmm(_t(X), _t(Y)).numpy(), **tol)
    Yt = np.swapaxes(Y, 1, 2).copy()
    assert np.allclose(gx.OPS["BatchMatMul"](_N(adj_x=False, adj_y=True), X, Yt), torch.bmm(_t(X), _t(Y)).numpy(), **tol)
    R = rng.randn(6, 5, 4).astype(dtype)
    for op, fn in (("Sum", torch.sum), ("Mean", torch.mean)):
        for ax in ([0], [1, 2], [2]):
            for keep in (False, True):
                got = gx.OPS[op](_N(keep_dims=keep), R, np.array(ax, np.int32))
                assert np.allclose(got, fn(_t(R), dim=ax, keepdim=keep).numpy(), **tol), (op, ax, keep)
    assert np.allclose(gx.OPS["Prod"](_N(keep_dims=False), R[:2, :2], np.array([2], np.int32)),
                       torch.prod(_t(R[:2, :2]), dim=2).numpy(), **tol)
    m = rng.rand(5, 4) < 0.2
    assert np.array_equal(gx.OPS["Any"](_N(keep_dims=False), m, np.array([1], np.int32)), torch.any(_t(m), dim=1).numpy())
    # AddN: left to right -- identical to a python-level chain of torch adds
    xs = [rng.randn(50).astype(dtype) * 10.0 ** k for k in (0, 6, -3, 6, 2)]
    acc = _t(xs[0])
    for x in xs[1:]:
        acc = acc + _t(x)
    assert np.array_equal(gx.OPS["AddN"](N0, *xs), acc.numpy())
    # UnsortedSegmentSum: one accumulator per segment, terms in index order (torch index_add_ on CPU is sequential)
    ids = rng.randint(0, 17, size=400).astype(np.int32)
    data = (rng.randn(400) * 10.0 ** rng.randint(-3, 7, size=400)).astype(dtype)
    want = torch.zeros(17, dtype=_t(data).dtype).index_add_(0, _t(ids.astype(np.int64)), _t(data))
    assert np.array_equal(gx.OPS["UnsortedSegmentSum"](N0, data, ids, np.int32(17)), want.numpy())
    data2 = rng.randn(40, 3).astype(dtype)
    ids2 = rng.randint(0, 5, size=40).astype(np.int32)
    want2 = torch.zeros(5, 3, dtype=_t(data2).dtype).index_add_(0, _t(ids2.astype(np.int64)), _t(data2))
    assert np.array_equal(gx.OPS["UnsortedSegmentSum"](N0, data2, ids2, np.int32(5)), want2.numpy())
    # LinSpace: start + step * i in T (sequence_ops.cc); torch fills symmetrically from both ends -> ulp-level
    for n in (1, 2, 28, 50, 128):
        got = gx.OPS["LinSpace"](N0, dtype(-1.0), dtype(1.0), np.int32(n))
        assert got.dtype == dtype and got.shape == (n,) and got[0] == -1.0
        if n > 1:
            _close(got, torch.linspace(-1.0, 1.0, n, dtype=_t(got).dtype).numpy(), dtype, ulps=4, atol=2 * np.finfo(dtype).eps)


def test_selection_and_layout_kernels():
    rng = np.random.RandomState(6)
    t, e = rng.randn(6, 4).astype(np.float32), rng.randn(6, 4).astype(np.float32)
    c1 = rng.rand(6) < 0.5                       # TF Select: a vector condition picks ROWS
    assert np.array_equal(gx.OPS["Select"](N0, c1, t, e), torch.where(_t(c1)[:, None], _t(t), _t(e)).numpy())
    c2 = rng.rand(6, 4) < 0.5
    assert np.array_equal(gx.OPS["Select"](N0, c2, t, e), torch.where(_t(c2), _t(t), _t(e)).numpy())
    p = rng.randn(9, 3).astype(np.float32)
    i = rng.randint(0, 9, size=(4, 5)).astype(np.int32)
    assert np.array_equal(gx.OPS["Gather"](N0, p, i), _t(p)[_t(i.astype(np.int64))].numpy())
    xs = [rng.randn(2, k, 3).astype(np.float32) for k in (1, 4, 2)]
    assert np.array_equal(gx.OPS["ConcatV2"](N0, *xs, np.int32(1)), torch.cat([_t(x) for x in xs], 1).numpy())
    v = rng.randn(5, 12).astype(np.float32)
    for got, want in zip(gx.OPS["Split"](_N(num_split=4), np.int32(1), v), torch.split(_t(v), 3, dim=1)):
        assert np.array_equal(got, want.numpy())
    assert np.array_equal(gx.OPS["Pad"](N0, v, np.array([[1, 0], [2, 3]])), torch.nn.functional.pad(_t(v), (2, 3, 1, 0)).numpy())
    assert np.array_equal(gx.OPS["Tile"](N0, v, np.array([2, 3])), _t(v).repeat(2, 3).numpy())
    w = rng.randn(2, 3, 4).astype(np.float32)
    assert np.array_equal(gx.OPS["Transpose"](N0, w, np.array([2, 0, 1])), _t(w).permute(2, 0, 1).numpy())
    assert np.array_equal(gx.OPS["Pack"](_N(axis=1), v, v + 1), torch.stack([_t(v), _t(v + 1)], 1).numpy())
    for got, want in zip(gx.OPS["Unpack"](_N(axis=1), w), torch.unbind(_t(w), 1)):
        assert np.array_equal(got, want.numpy())
    assert np.array_equal(gx.OPS["Slice"](N0, w, np.array([0, 1, 1]), np.array([2, -1, 2])), _t(w)[0:2, 1:, 1:3].numpy())
    assert np.array_equal(gx.OPS["ExpandDims"](N0, v, np.int32(1)), _t(v).unsqueeze(1).numpy())
    a0, a1 = gx.OPS["BroadcastGradientArgs"](N0, np.array([64, 1]), np.array([64, 2500]))
    assert list(a0) == [1] and list(a1) == []
    a0, a1 = gx.OPS["BroadcastGradientArgs"](N0, np.array([64, 50]), np.array([50]))
    assert list(a0) == [] and list(a1) == [0]


@pytest.mark.parametrize("dtype", DT)
def test_apply_adam_matches_a_torch_restatement_and_torch_optim(dtype):
    """ApplyAdam (training_ops.cc): TF's epsilon sits OUTSIDE the bias correction ("epsilon hat"), torch.optim.Adam's
    inside -- the two agree to O(eps / sqrt(v)); a torch-op restatement of TF's formula must agree to rounding."""
    rng = np.random.RandomState(7)
    T = dtype
    var, g = rng.randn(500).astype(T), (rng.randn(500) * 3).astype(T)
    m, v = np.zeros(500, T), np.zeros(500, T)
    lr, b1, b2, eps = T(1e-4), T(0.9), T(0.999), T(1e-8)
    tv = torch.nn.Parameter(_t(var.copy()))
    opt = torch.optim.Adam([tv], lr=float(lr), betas=(0.9, 0.999), eps=1e-8)
    tm, tvv, tvar = _t(m.copy()), _t(v.copy()), _t(var.copy())
    for step in range(1, 4):
        b1p, b2p = T(b1 ** step), T(b2 ** step)
        var, m, v = gx.apply_adam(var, m, v, b1p, b2p, lr, b1, b2, eps, g)
        # restatement with torch ops
        tg = _t(g)
        alpha = float(lr) * torch.sqrt(torch.tensor(1 - float(b2p), dtype=tg.dtype)) / (1 - float(b1p))
        tm = tm + (tg - tm) * (1 - float(b1))
        tvv = tvv + (tg * tg - tvv) * (1 - float(b2))
        tvar = tvar - (tm * alpha) / (torch.sqrt(tvv) + float(eps))
        assert np.allclose(var, tvar.numpy(), rtol=0, atol=8 * np.finfo(T).eps * max(1.0, np.abs(var).max()))
        assert np.allclose(m, tm.numpy(), rtol=8 * np.finfo(T).eps) and np.allclose(v, tvv.numpy(), rtol=8 * np.finfo(T).eps)
        tv.grad = _t(g.copy())
        opt.step()
        # same trajectory as torch.optim.Adam up to the epsilon placement (|g| ~ 3 >> eps) and fp rounding
        assert np.abs(var - tv.detach().numpy()).max() <= 2e-3 * float(lr) * step
        g = (g * T(0.7) + rng.randn(500).astype(T)).astype(T)


def test_cast_semantics():
    """Cast float -> int32 truncates toward zero (TF and torch alike); bool -> float gives 0/1."""
    x = np.array([-2.7, -0.5, 0.5, 2.7, 49.999], np.float32)
    assert np.array_equal(x.astype(np.int32), _t(x).to(torch.int32).numpy())
    b = np.array([True, False])
    assert np.array_equal(b.astype(np.float32), _t(b).to(torch.float32).numpy())
