"""CPU tests of the oracle restatement (SURVEY appendix D known-answer tests +
golden snapshot).  No GPU, no product code."""
import os

import numpy as np
import pytest
import torch

from oracle import air_oracle as ao
from oracle import air_oracle_torch as at
from oracle.synth import blob_canvases

HP = dict(ao.TRAINING_HP)


def _setup(B=4, seed=3):
    images, targets = blob_canvases(B, 50, 2, seed=seed)
    return images, targets, ao.init_params(HP, 0), ao.make_noise(HP, B, 1)


def test_golden_snapshot(golden_dir):
    """The committed fixture is reproduced bit-for-bit in structure and to
    fp32 round-off in value (BLAS summation order may differ across hosts)."""
    g = np.load(os.path.join(golden_dir, "air_b4.npz"))
    images, targets, params, noise = _setup()
    lo0 = ao.annealed_value(ao.TRAINING_ANNEALING["z_pres_prior_log_odds"], 0)
    for tag, train, lo in (("train_lo9", True, lo0), ("train_lom2", True, -2.0), ("test_lom2", False, -2.0)):
        o = ao.air_forward(params, images, targets, noise, HP, train, lo)
        assert np.array_equal(o["rec_num_digits"], g[tag + "/rec_num_digits"])
        np.testing.assert_allclose(o["reconstruction"], g[tag + "/reconstruction"], atol=2e-5)
        for k in ("rec_scales", "rec_shifts", "rec_windows", "rec_latents", "z_pres_probs"):
            np.testing.assert_allclose(o[k], g[tag + "/" + k], atol=2e-5, rtol=1e-4)
        for k in ("z_pres_kls", "scale_kls", "shift_kls", "vae_kls"):
            np.testing.assert_allclose(o[k], g[tag + "/" + k], rtol=1e-4, atol=1e-4)
        # ELBO: OOB residues pass through log(r + 1e-9) -> loose
        np.testing.assert_allclose(o["loss"], g[tag + "/loss"], rtol=1e-2)


def test_fp32_vs_fp64_error_bars():
    images, targets, params, noise = _setup(8, 5)
    o32 = ao.air_forward(params, images, targets, noise, HP, True, -2.0)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    n64 = {k: v.astype(np.float64) for k, v in noise.items()}
    o64 = ao.air_forward(p64, images.astype(np.float64), targets, n64, HP, True, -2.0)
    assert np.abs(o32["reconstruction"] - o64["reconstruction"]).max() < 1e-5
    assert np.array_equal(o32["rec_num_digits"], o64["rec_num_digits"])
    assert abs(o32["loss"] - o64["loss"]) / abs(o64["loss"]) < 3e-2
    # BCE evaluated on the SAME reconstruction agrees tightly
    r = o32["reconstruction"].astype(np.float64)
    x = images.astype(np.float64)
    bce = -np.sum(x * np.log(r + ao.EPS) + (1 - x) * np.log(1 - r + ao.EPS), axis=1)
    np.testing.assert_allclose(o32["reconstruction_loss"], bce, rtol=1e-5)


def test_torch_twin_matches_numpy():
    images, targets, params, noise = _setup(6, 7)
    for train in (True, False):
        o = ao.air_forward(params, images, targets, noise, HP, train, -2.0)
        ot = at.air_forward(at.to_torch(params), torch.tensor(images), torch.tensor(targets),
                            at.to_torch(noise), HP, train, -2.0)
        assert np.array_equal(o["rec_num_digits"], ot["rec_num_digits"].numpy())
        assert np.abs(o["reconstruction"] - ot["reconstruction"].numpy()).max() < 1e-5
        for k in ("z_pres_kls", "scale_kls", "shift_kls", "vae_kls"):
            np.testing.assert_allclose(o[k], ot[k].numpy(), rtol=1e-4, atol=1e-4)


# ---- sampler KATs (appendix D) --------------------------------------------

def test_sampler_far_outside_is_zero():
    rng = np.random.RandomState(0)
    U = rng.uniform(0, 1, (3, 50, 50)).astype(np.float32)
    theta = np.tile(np.array([[0.3, 0, 5.0], [0, 0.3, -4.0]], np.float32), (3, 1, 1))
    out = ao.transformer(U, theta, (28, 28))
    assert np.abs(out).max() < 4e-6
    out64 = ao.transformer(U.astype(np.float64), theta.astype(np.float64), (28, 28))
    assert np.abs(out64).max() < 1e-12


def test_sampler_identity_is_close_not_exact():
    rng = np.random.RandomState(1)
    U = rng.uniform(0, 1, (2, 50, 50)).astype(np.float32)
    theta = np.tile(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (2, 1, 1))
    out = ao.transformer(U, theta, (50, 50))
    err = np.abs(out - U).max()
    assert 0 < err < 2e-3


def _interp_matrix(n_in, n_out, a, b, dtype=np.float64):
    """Separable form: rows of R hold the (<=2) tap weights of one output coord."""
    t = np.linspace(-1, 1, n_out).astype(dtype)
    X = ((a * t + b) + 1.0) * (n_in - 1.001) / 2.0
    x0 = np.floor(X).astype(np.int64)
    x1 = x0 + 1
    x0c, x1c = np.clip(x0, 0, n_in - 1), np.clip(x1, 0, n_in - 1)
    R = np.zeros((n_out, n_in), dtype)
    for j in range(n_out):
        R[j, x0c[j]] += x1c[j] - X[j]
        R[j, x1c[j]] += X[j] - x0c[j]
    return R


def test_sampler_is_separable_and_adjoint():
    rng = np.random.RandomState(2)
    U = rng.uniform(0, 1, (1, 28, 28))
    s, x, y = 0.45, 0.2, -0.3
    theta = np.array([[[1 / s, 0, -x / s], [0, 1 / s, -y / s]]])
    out = ao.transformer(U, theta, (50, 50))[0]
    Ry = _interp_matrix(28, 50, 1 / s, -y / s)
    Rx = _interp_matrix(28, 50, 1 / s, -x / s)
    sep = Ry @ U[0] @ Rx.T
    assert np.abs(out - sep).max() < 1e-12
    V = rng.uniform(-1, 1, (50, 50))
    lhs = np.sum(sep * V)
    rhs = np.sum(U[0] * (Ry.T @ V @ Rx))
    assert abs(lhs - rhs) < 1e-9


def test_sampler_gradients_finite_difference():
    rng = np.random.RandomState(3)
    U = torch.tensor(rng.uniform(0, 1, (2, 50, 50)), dtype=torch.float64)
    sxy = torch.tensor([[0.41, 0.13, -0.22], [0.62, -0.35, 0.08]], dtype=torch.float64, requires_grad=True)
    G = torch.tensor(rng.uniform(-1, 1, (2, 28, 28)), dtype=torch.float64)

    def f(p):
        z = torch.zeros_like(p[:, 0])
        th = torch.stack([torch.stack([p[:, 0], z, p[:, 1]], 1), torch.stack([z, p[:, 0], p[:, 2]], 1)], 1)
        return (at.transformer(U, th, (28, 28)) * G).sum()

    (g,) = torch.autograd.grad(f(sxy), sxy)
    h = 1e-7
    for b in range(2):
        for k in range(3):
            d = torch.zeros_like(sxy)
            d[b, k] = h
            fd = (f(sxy.detach() + d) - f(sxy.detach() - d)) / (2 * h)
            assert abs(fd - g[b, k]) < 1e-5 * max(1.0, abs(g[b, k])), (b, k, fd, g[b, k])


# ---- pointwise KATs ---------------------------------------------------------

def test_concrete_kl_zero_when_prior_equals_posterior():
    y = np.linspace(-5, 5, 11).astype(np.float32)
    lo = np.full(11, 0.7, np.float32)
    kl = ao.concrete_binary_kl_mc_sample(y, np.float32(0.7), 1.0, lo, 1.0)
    assert np.all(kl == 0)


def test_gauss_kl_zero_at_prior():
    mean = np.full((3, 2), -1.0, np.float32)
    lv = np.full((3, 2), np.log(np.float32(0.05)), np.float32)
    kl = ao._gauss_kl(np.log(np.float32(0.05)), lv, np.exp(lv), 0.05, mean, -1.0)
    assert np.abs(kl).max() < 1e-6


def test_lstm_zero_weights():
    x = np.random.RandomState(0).uniform(0, 1, (3, 20)).astype(np.float32)
    c = np.zeros((3, 4), np.float32)
    h = np.zeros((3, 4), np.float32)
    c2, h2 = ao.lstm_cell(x, c, h, np.zeros((24, 16), np.float32), np.zeros(16, np.float32))
    assert np.all(c2 == 0) and np.all(h2 == 0)


def test_lstm_gate_order_and_forget_bias():
    # only gate j (2nd block) and i (1st) open -> c' = sigmoid(i)*tanh(j)
    R = 2
    kernel = np.zeros((1 + R, 4 * R), np.float32)
    bias = np.zeros(4 * R, np.float32)
    bias[0:R] = 100.0       # i -> 1
    bias[R:2 * R] = 0.5     # j
    bias[2 * R:3 * R] = -1.0  # f + forget_bias(1.0) = 0 -> sigmoid = .5
    bias[3 * R:] = 100.0    # o -> 1
    c = np.full((1, R), 2.0, np.float32)
    c2, h2 = ao.lstm_cell(np.zeros((1, 1), np.float32), c, np.zeros((1, R), np.float32), kernel, bias)
    np.testing.assert_allclose(c2, 2.0 * 0.5 + np.tanh(0.5), rtol=1e-6)
    np.testing.assert_allclose(h2, np.tanh(c2), rtol=1e-6)


def test_annealing_schedule():
    sch = ao.TRAINING_ANNEALING["z_pres_prior_log_odds"]
    assert abs(ao.annealed_value(sch, 0) - 9.21034) < 1e-4
    assert abs(ao.annealed_value(sch, 3000) - 6.90776) < 1e-4
    assert abs(ao.annealed_value(sch, 39000) - (-20.0301)) < 2e-3
    assert abs(ao.annealed_value(sch, 200000) - (-20.0301)) < 2e-3


def test_stop_logic_test_mode():
    """z=(1,1,0) -> digits 2 and the 3rd write is masked; z=(0,.,.) -> digits 0,
    R == 0 and the step-1 z_pres KL is still counted (old-S mask)."""
    images, targets, params, _ = _setup(2, 11)
    noise = ao.make_noise(HP, 2, 1)
    # force z via the uniform sample: u->1 gives z=1, u->0 gives z=0
    noise["u"][:, 0] = [1 - 1e-7, 1 - 1e-7, 1e-7]
    noise["u"][:, 1] = [1e-7, 0.5, 0.5]
    o = ao.air_forward(params, images, targets, noise, HP, False, -2.0)
    assert list(o["rec_num_digits"]) == [2, 0]
    assert np.all(o["_running_recon"][1] == 0)
    two = o["_z_pres"][0, 0] * o["_window_recon"][0, 0] + o["_z_pres"][0, 1] * o["_window_recon"][0, 1]
    np.testing.assert_allclose(o["_running_recon"][0], two, atol=1e-6)
    # item 1: only the first z KL was added to the running loss
    np.testing.assert_allclose(o["_running_loss"][1], o["z_pres_kls"][1, 0], rtol=1e-6)


def test_fixed_n_equals_early_exit():
    images, targets, params, noise = _setup(4, 13)
    noise["u"][:] = 1e-7          # everything stops after step 1
    a = ao.air_forward(params, images, targets, noise, HP, False, -2.0, early_exit=True)
    b = ao.air_forward(params, images, targets, noise, HP, False, -2.0, early_exit=False)
    assert a["steps_executed"] == 1 and b["steps_executed"] == 3
    assert a["rec_scales"].shape[1] == 1 and b["rec_scales"].shape[1] == 3
    assert np.array_equal(a["loss_per_item"], b["loss_per_item"])
    assert np.array_equal(a["reconstruction"], b["reconstruction"])
    assert np.array_equal(a["rec_num_digits"], b["rec_num_digits"])


# ---- optimizer KATs ---------------------------------------------------------

def test_adam_closed_form_and_differs_from_torch():
    p = {"w": np.array([1.0, -2.0], np.float32)}
    g = {"w": np.array([1e-6, 3.0], np.float32)}
    m = {"w": np.zeros(2, np.float32)}
    v = {"w": np.zeros(2, np.float32)}
    lr, b1, b2, eps = 1e-4, 0.9, 0.999, 1e-8
    p2, m2, v2 = ao.adam_step(dict(p), g, m, v, 1, lr)
    lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
    expect = p["w"] - lr_t * ((1 - b1) * g["w"]) / (np.sqrt((1 - b2) * g["w"] ** 2) + eps)
    np.testing.assert_allclose(p2["w"], expect, rtol=1e-6)
    w = torch.tensor(p["w"].copy(), requires_grad=True)
    opt = torch.optim.Adam([w], lr=lr, betas=(b1, b2), eps=eps)
    w.grad = torch.tensor(g["w"])
    opt.step()
    # tiny gradient: epsilon placement matters -> TF-Adam != torch Adam
    assert abs(float(w.detach()[0]) - float(p2["w"][0])) > 1e-6 * lr


def test_clip_by_global_norm():
    g = {"a": np.array([3.0], np.float32), "b": np.array([4.0], np.float32)}
    c, gn = ao.clip_by_global_norm(g, 1.0)
    assert abs(gn - 5.0) < 1e-6
    np.testing.assert_allclose([c["a"][0], c["b"][0]], [0.6, 0.8], rtol=1e-6)
    c, _ = ao.clip_by_global_norm(g, 100.0)
    np.testing.assert_allclose([c["a"][0], c["b"][0]], [3.0, 4.0], rtol=1e-6)


def test_dp_equivalence_of_mean_gradients():
    """grad of mean over 1 x 2B == mean of grads of 2 shards of B."""
    images, targets, params, noise = _setup(4, 17)
    pt = at.to_torch(params, dtype=torch.float64, requires_grad=True)
    nt = at.to_torch(noise, dtype=torch.float64)
    im, tg = torch.tensor(images, dtype=torch.float64), torch.tensor(targets)
    _, gfull = at.loss_and_grads(pt, im, tg, nt, HP, -2.0)
    gfull = {k: v.clone() for k, v in gfull.items()}
    acc = None
    for sl in (slice(0, 2), slice(2, 4)):
        nsl = {k: v[:, sl] for k, v in nt.items()}
        _, gs = at.loss_and_grads(pt, im[sl], tg[sl], nsl, HP, -2.0)
        acc = {k: v.clone() for k, v in gs.items()} if acc is None else {k: acc[k] + gs[k] for k in gs}
    for k in gfull:
        assert torch.allclose(gfull[k], acc[k] / 2, rtol=1e-9, atol=1e-12), k
