"""CPU oracle for the AIR hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``tf-attend-infer-repeat_amd/``)
never imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED in the brief's sense: no golden vector held by the reference's own tests and no output of the reference
RUN here exists (neither can: see below).  What stands in for them, and how far it reaches:

PARITY PIN: the reference (aakhundov/tf-attend-infer-repeat) ships no tests, golden
vectors or fixtures for this path, and its arithmetic lives in TensorFlow 1.3.0
(un-vendored, not installable here: no cp310 wheel, no network) -- so no output
of a RUNNING reference exists.  What does exist is the reference's own
serialized training graph, ``model/air-model.meta`` (written by TF 1.3 after 270k
iterations).  ``oracle/graphdef_exec.py`` executes that graph node by node in numpy
(forward while-loop, loss, the whole tf.gradients backward, clip, ApplyAdam) and
``tests/test_graph_exec.py`` asserts that this restatement is BIT-IDENTICAL to it in
fp32 on every fetched output (train model B=64, test model at dynamic B, annealing,
early loop exit), and equal to <= 2e-4 / 1e-9 on all 36 gradients / the Adam update
in fp64.  ``tests/golden/graph_b64.npz`` holds vectors produced by that graph
execution.  The pin is to the graph's dataflow and constants with numpy's
elementary kernels -- not to the last-bit rounding of TF's Eigen kernels, which
cannot be run here.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  All arithmetic is done in ``dtype`` (np.float32 mirrors the
reference bit-for-bit in op order -- one rounding per TF op, no FMA; np.float64
is the twin used for error bars).  All randomness is an explicit input
(``noise`` dict) because every RNG op in the reference graph is unseeded.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

EPS = 10e-10  # the sources write 10e-10 (= 1e-9): air_model.py:95,587-588; concrete.py:20,33


# --------------------------------------------------------------------------- #
# hyper-parameters
# --------------------------------------------------------------------------- #

# AIRModel.__init__ defaults, air/air_model.py:13-22
DEFAULT_HP = dict(
    max_steps=3, max_digits=2, rnn_units=256, canvas_size=50, windows_size=28,
    vae_latent_dimensions=50, vae_recognition_units=(512, 256), vae_generative_units=(256, 512),
    scale_prior_mean=-1.0, scale_prior_variance=0.1, shift_prior_mean=0.0, shift_prior_variance=1.0,
    vae_prior_mean=0.0, vae_prior_variance=1.0, vae_likelihood_std=0.3,
    scale_hidden_units=64, shift_hidden_units=64, z_pres_hidden_units=64,
    z_pres_prior_log_odds=-2.0, z_pres_temperature=1.0, stopping_threshold=0.99,
    learning_rate=1e-3, gradient_clipping_norm=100.0,
)

# what training.py:100-122 actually passes (the benchmark configuration)
TRAINING_HP = dict(
    DEFAULT_HP,
    scale_prior_variance=0.05, z_pres_prior_log_odds=-0.01,
    learning_rate=1e-4, gradient_clipping_norm=1.0,
)

# training.py:110-116
TRAINING_ANNEALING = {
    "z_pres_prior_log_odds": {
        "init": 10000.0, "min": 0.000000001, "factor": 0.1, "iters": 3000,
        "staircase": False, "log": True,
    }
}


def annealed_value(schedule, global_step, dtype=np.float32):
    """air_model.py:94-121 (_create_annealed_tensor) on top of
    tf.train.exponential_decay: init * factor ** (step / iters)."""
    f = np.dtype(dtype).type
    p = f(global_step) / f(schedule["iters"])
    if schedule.get("staircase", False):
        p = np.floor(p)
    value = f(schedule["init"]) * np.power(f(schedule["factor"]), p, dtype=dtype)
    if "min" in schedule:
        value = np.maximum(value, f(schedule["min"]))
    if "max" in schedule:
        value = np.minimum(value, f(schedule["max"]))
    if schedule.get("log", False):
        value = np.log(value + f(EPS))
    return f(value)


# --------------------------------------------------------------------------- #
# parameters (names/shapes: model/air-model.index)
# --------------------------------------------------------------------------- #

def param_shapes(hp):
    D = hp["canvas_size"] ** 2
    d = hp["windows_size"] ** 2
    R = hp["rnn_units"]
    Z = hp["vae_latent_dimensions"]
    shapes = OrderedDict()
    shapes["rnn/kernel"] = (D + R, 4 * R)
    shapes["rnn/bias"] = (4 * R,)
    for head, hid, k in (("scale/mean", hp["scale_hidden_units"], 1),
                         ("scale/log_variance", hp["scale_hidden_units"], 1),
                         ("shift/mean", hp["shift_hidden_units"], 2),
                         ("shift/log_variance", hp["shift_hidden_units"], 2),
                         ("z_pres/log_odds", hp["z_pres_hidden_units"], 1)):
        shapes[head + "/hidden/weights"] = (R, hid)
        shapes[head + "/hidden/biases"] = (hid,)
        shapes[head + "/output/weights"] = (hid, k)
        shapes[head + "/output/biases"] = (k,)
    prev = d
    for i, u in enumerate(hp["vae_recognition_units"]):
        shapes["vae/recognition_%d/weights" % (i + 1)] = (prev, u)
        shapes["vae/recognition_%d/biases" % (i + 1)] = (u,)
        prev = u
    shapes["vae/rec_mean/weights"] = (prev, Z)
    shapes["vae/rec_mean/biases"] = (Z,)
    shapes["vae/rec_log_variance/weights"] = (prev, Z)
    shapes["vae/rec_log_variance/biases"] = (Z,)
    prev = Z
    for i, u in enumerate(hp["vae_generative_units"]):
        shapes["vae/generative_%d/weights" % (i + 1)] = (prev, u)
        shapes["vae/generative_%d/biases" % (i + 1)] = (u,)
        prev = u
    shapes["vae/gen_mean/weights"] = (prev, d)
    shapes["vae/gen_mean/biases"] = (d,)
    return shapes


def init_params(hp, seed=0, dtype=np.float32):
    """Glorot/Xavier-uniform weights, zero biases -- the TF1.3 defaults of
    BasicLSTMCell (get_variable default initializer) and
    layers.fully_connected (xavier_initializer, zeros); limits
    sqrt(6/(fan_in+fan_out)) verified against air-model.meta initializer consts."""
    rng = np.random.RandomState(seed)
    params = OrderedDict()
    for name, shape in param_shapes(hp).items():
        if len(shape) == 2:
            limit = math.sqrt(6.0 / (shape[0] + shape[1]))
            params[name] = rng.uniform(-limit, limit, size=shape).astype(dtype)
        else:
            params[name] = np.zeros(shape, dtype=dtype)
    return params


def make_noise(hp, batch, seed=0, dtype=np.float32):
    """Noise protocol: per step, in graph order: scale, shift,
    vae-z, vae-x normals (air_model.py:127, vae.py:23,37) and the Concrete
    uniform (concrete.py:23)."""
    rng = np.random.RandomState(seed)
    N = hp["max_steps"]
    Z = hp["vae_latent_dimensions"]
    d = hp["windows_size"] ** 2
    noise = dict(
        eps_scale=rng.standard_normal((N, batch, 1)),
        eps_shift=rng.standard_normal((N, batch, 2)),
        eps_z=rng.standard_normal((N, batch, Z)),
        eps_x=rng.standard_normal((N, batch, d)),
        u=rng.uniform(0.0, 1.0, size=(N, batch)),
    )
    return {k: v.astype(dtype) for k, v in noise.items()}


# --------------------------------------------------------------------------- #
# elementary TF ops
# --------------------------------------------------------------------------- #

def sigmoid(x):
    # tf.nn.sigmoid == 1 / (1 + exp(-x)) (Eigen scalar_sigmoid_op)
    one = x.dtype.type(1.0)
    return one / (one + np.exp(-x))


def softplus(x):
    # tf.nn.softplus (TF 1.3 softplus_op.h functor): threshold = log(eps) + 2;
    # x > -threshold -> x;  x < threshold -> exp(x);  else log(exp(x) + 1).
    one = x.dtype.type(1.0)
    threshold = x.dtype.type(np.log(np.finfo(x.dtype).eps) + 2.0)
    with np.errstate(over="ignore"):
        mid = np.log(np.exp(x) + one)
    return np.where(x > -threshold, x, np.where(x < threshold, np.exp(x), mid)).astype(x.dtype)


def fully_connected(x, W, b, act=None):
    """tf.contrib.layers.fully_connected: act(x @ W + b); default act in the
    reference call sites is ReLU (air_model.py:292, 297, 309, 314, 374)."""
    y = x @ W + b
    if act == "relu":
        y = np.maximum(y, y.dtype.type(0.0))
    elif act == "softplus":
        y = softplus(y)
    return y


# --------------------------------------------------------------------------- #
# spatial transformer -- air/transformer.py
# --------------------------------------------------------------------------- #

def _linspace(n, dtype):
    # tf.linspace(-1, 1, n): start + step * i evaluated in T (transformer.py:126-129)
    f = np.dtype(dtype).type
    if n == 1:
        return np.array([-1.0], dtype=dtype)
    step = f(f(1.0) - f(-1.0)) / f(n - 1)
    return (f(-1.0) + step * np.arange(n, dtype=dtype)).astype(dtype)


def meshgrid(height, width, dtype=np.float32):
    """transformer.py:119-136 (_meshgrid): returns x_t, y_t flattened row-major."""
    x_lin = _linspace(width, dtype)
    y_lin = _linspace(height, dtype)
    x_t = np.ones((height, 1), dtype=dtype) @ x_lin[None, :]
    y_t = y_lin[:, None] @ np.ones((1, width), dtype=dtype)
    return x_t.reshape(-1), y_t.reshape(-1)


def transformer(U, theta, out_size, return_aux=False):
    """transformer.py:18-175.  U [B,Hi,Wi] (single channel), theta [B,2,3],
    out_size (Ho, Wo) -> [B,Ho,Wo].

    Follows _transform (:138-171) then _interpolate (:56-117): scale by
    (W - 1.001)/2, floor, +1, THEN clip the integer corners, gather the four
    taps, weights from the CLIPPED corners, add_n in the order a, b, c, d."""
    dtype = U.dtype
    f = dtype.type
    B, Hi, Wi = U.shape
    Ho, Wo = out_size
    x_t, y_t = meshgrid(Ho, Wo, dtype)

    # T_g = matmul(theta, grid) (:159); grid rows are (x_t, y_t, 1)
    th = theta.astype(dtype)
    x_s = (th[:, 0, 0:1] * x_t[None, :] + th[:, 0, 1:2] * y_t[None, :]) + th[:, 0, 2:3] * f(1.0)
    y_s = (th[:, 1, 0:1] * x_t[None, :] + th[:, 1, 1:2] * y_t[None, :]) + th[:, 1, 2:3] * f(1.0)

    # _interpolate (:75-76)
    width_f, height_f = f(Wi), f(Hi)
    x = (x_s + f(1.0)) * (width_f - f(1.001)) / f(2.0)
    y = (y_s + f(1.0)) * (height_f - f(1.001)) / f(2.0)

    # (:79-87)
    x0 = np.floor(x).astype(np.int32)
    x1 = x0 + 1
    y0 = np.floor(y).astype(np.int32)
    y1 = y0 + 1
    x0 = np.clip(x0, 0, Wi - 1)
    x1 = np.clip(x1, 0, Wi - 1)
    y0 = np.clip(y0, 0, Hi - 1)
    y1 = np.clip(y1, 0, Hi - 1)

    # (:88-105) gathers from the flat image
    bidx = np.arange(B)[:, None]
    Ia = U[bidx, y0, x0]
    Ib = U[bidx, y1, x0]
    Ic = U[bidx, y0, x1]
    Id = U[bidx, y1, x1]

    # (:108-116)
    x0_f, x1_f = x0.astype(dtype), x1.astype(dtype)
    y0_f, y1_f = y0.astype(dtype), y1.astype(dtype)
    wa = (x1_f - x) * (y1_f - y)
    wb = (x1_f - x) * (y - y0_f)
    wc = (x - x0_f) * (y1_f - y)
    wd = (x - x0_f) * (y - y0_f)
    out = ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id  # tf.add_n, assumed left-to-right
    out = out.reshape(B, Ho, Wo)
    if return_aux:
        return out, dict(x=x, y=y, x0=x0, x1=x1, y0=y0, y1=y1)
    return out


CARRIED_CHUNKS = 16     # at most this many chunks per (input pixel, tap) term stream of order="carried16" ...
CARRIED_CHUNK_MIN = 64  # ... none of them shorter than this (a stream of up to 64 terms is one chunk)


def _seq_sum(start, terms):
    """fl(...fl(fl(start + t0) + t1)... ) in the dtype of `terms` (np.add.accumulate adds left to right)"""
    if len(terms) == 0:
        return terms.dtype.type(start)
    return np.add.accumulate(np.concatenate([[start], terms]).astype(terms.dtype))[-1]


def carried_segment_sum(ids4, vals4, num, chunks=CARRIED_CHUNKS, chunk_min=CARRIED_CHUNK_MIN):
    """The accumulation order of backward="reference_carried" (order="carried16"): the four Gather gradients ids4 / vals4
    (taps a, b, c, d) into `num` accumulators.

    A slot (input pixel) whose four term streams all have at most `chunk_min` terms is summed exactly as the reference's
    UnsortedSegmentSum sums it: one accumulator through its a-, b-, c-, d-terms.  That is every slot but the ones that collect
    the out-of-range canvas (the four window corners; at canvases above 64 pixels the border rows / columns as well).
    A slot with a longer stream keeps the reference's left-to-right structure AND the magnitude of its roundings, on chunks
    that can be walked side by side.  Every tap's stream of n terms is cut into contiguous chunks of
    cs = max(ceil(n / chunks), chunk_min) terms; over all chunks of the slot in stream order (a, b, c, d), with P_0 = +0.0:
        C_k = chunk k summed sequentially from +0.0
        Q_k = chunk k summed sequentially from P_k                   (the chain the reference runs there: P_k stands for
                                                                      the reference's accumulator at the chunk's start,
                                                                      so every add rounds at the magnitude it rounds at
                                                                      in the sequential order)
        P_k+1 = P_k + C_k
    and the slot's sum is  Q_last + sum_{k < last} (Q_k - P_k+1),  the corrections added left to right from +0.0.  Q_k and
    P_k+1 are two roundings of the same real number -- the accumulator after chunk k -- so their difference is a few ulps
    and EXACT (Sterbenz); the sum is the telescoped form of sum_k (Q_k - P_k), evaluated without the two roundings per
    chunk at full magnitude that the plain form costs (2116 corner streams at initialisation: mean |error| 4.37e3 against
    the sequential order's 4.34e3; the plain form 6.3e3, the C_k alone added left to right 8.8e3 with a 5.7x heavier tail --
    tests/test_graph_exec.py).
    All C of a slot are independent, P is their running sum, all Q are independent given P: two chains of n / chunks adds
    instead of one of 4 n.  The out-of-range terms cancel pairwise in exact arithmetic (a against c, b against d); what is
    left of them is rounding residue, and its size is set by the magnitude of the accumulator the terms are added to --
    which summing every chunk from +0.0 changes and this order keeps."""
    dtype = vals4[0].dtype
    f = dtype.type
    per_tap = []
    counts = np.zeros((4, num), np.int64)
    for k in range(4):
        order = np.argsort(ids4[k], kind="stable")
        si, sv = ids4[k][order], vals4[k][order]
        counts[k] = np.bincount(si, minlength=num)
        per_tap.append((si, sv, np.concatenate([[0], np.cumsum(counts[k])])))
    long_slot = (counts > chunk_min).any(0)
    out = np.zeros(num, dtype)
    keep = [~long_slot[i] for i in ids4]
    np.add.at(out, np.concatenate([i[m] for i, m in zip(ids4, keep)]), np.concatenate([v[m] for v, m in zip(vals4, keep)]))
    for slot in np.flatnonzero(long_slot):
        P, corr, Q = f(0), f(0), None
        for k in range(4):
            si, sv, off = per_tap[k]
            t = sv[off[slot]:off[slot + 1]]
            n = len(t)
            if n == 0:
                continue
            cs = max(-(-n // chunks), chunk_min)
            for k0 in range(0, n, cs):
                ch = t[k0:k0 + cs]
                if Q is not None:
                    corr = f(corr + f(Q - P))                     # the previous chunk's Q against the prefix behind it
                Q = _seq_sum(P, ch)
                P = f(P + _seq_sum(f(0), ch))
        out[slot] = f(Q + corr)
    return out


def transformer_backward(U, theta, out_size, d_out, order="sequential"):
    """What tf.gradients builds for transformer(U, theta, out_size) (transformer.py:56-171), in the op
    order of the reference's saved graph (model/air-model.meta, `.../st_backward/...` gradient nodes;
    executed by oracle/graphdef_exec.py, pinned to this function by tests/test_graph_exec.py):

      * d U: the four Gather gradients are concatenated (a, b, c, d) and reduced by ONE
        UnsortedSegmentSum -- np.add.at visits the terms in that order, like the TF CPU kernel
        (order="sequential", the default).
        order="carried16" (AIRModel(backward="reference_carried")): the reference's order for every slot with short streams,
        chunks walked from a carried estimate of the reference's accumulator for the long ones (carried_segment_sum);
      * coordinate gradients: mul_10..13_grad / mul_6..9_grad products, Sub negations, then AddN_10
        (x) and AddN_11 (y) over the legs of wa, wb, wc, wd left to right; truediv_grad, mul_grad;
      * d theta: MatMul_grad -- contraction of (d x_s, d y_s) with the grid rows (x_t, y_t, 1).
    Returns (d_U [B,Hi,Wi], d_theta [B,2,3])."""
    dtype = U.dtype
    f = dtype.type
    B, Hi, Wi = U.shape
    Ho, Wo = out_size
    _, aux = transformer(U, theta, out_size, return_aux=True)
    x, y, x0, x1, y0, y1 = (aux[k] for k in ("x", "y", "x0", "x1", "y0", "y1"))
    g = d_out.reshape(B, Ho * Wo).astype(dtype)
    x0_f, x1_f, y0_f, y1_f = (v.astype(dtype) for v in (x0, x1, y0, y1))
    wx0, wx1, wy0, wy1 = x1_f - x, x - x0_f, y1_f - y, y - y0_f
    bidx = np.arange(B)[:, None]
    Ia, Ib, Ic, Id = U[bidx, y0, x0], U[bidx, y1, x0], U[bidx, y0, x1], U[bidx, y1, x1]
    # d U: [a-terms of every output pixel, then b, c, d] into one accumulator per input pixel
    d_U = np.zeros((B, Hi * Wi), dtype)
    assert order in ("sequential", "carried16"), order
    for b in range(B):
        idx = [y0[b] * Wi + x0[b], y1[b] * Wi + x0[b], y0[b] * Wi + x1[b], y1[b] * Wi + x1[b]]
        val = [(wx0[b] * wy0[b]) * g[b], (wx0[b] * wy1[b]) * g[b], (wx1[b] * wy0[b]) * g[b], (wx1[b] * wy1[b]) * g[b]]
        if order == "carried16":
            d_U[b] = carried_segment_sum(idx, val, Hi * Wi)
            continue
        np.add.at(d_U[b], np.concatenate(idx), np.concatenate(val))
    ga, gb, gc, gd = g * Ia, g * Ib, g * Ic, g * Id
    dX = ((-(ga * wy0) + -(gb * wy1)) + gc * wy0) + gd * wy1
    dY = ((-(wx0 * ga) + wx0 * gb) + -(wx1 * gc)) + wx1 * gd
    dxs = (dX / f(2.0)) * (f(Wi) - f(1.001))
    dys = (dY / f(2.0)) * (f(Hi) - f(1.001))
    x_t, y_t = meshgrid(Ho, Wo, dtype)
    grid = np.stack([x_t, y_t, np.ones_like(x_t)])                       # [3, Ho*Wo]
    d_theta = np.stack([dxs @ grid.T, dys @ grid.T], axis=1)             # [B, 2, 3]
    return d_U.reshape(B, Hi, Wi), d_theta.astype(dtype)


# --------------------------------------------------------------------------- #
# Concrete / Gumbel-Softmax -- air/concrete.py
# --------------------------------------------------------------------------- #

def concrete_binary_pre_sigmoid_sample(log_odds, temperature, u):
    """concrete.py:20-27 with the uniform sample u injected."""
    f = log_odds.dtype.type
    noise = np.log(u + f(EPS)) - np.log(f(1.0) - u + f(EPS))
    return (log_odds + noise) / f(temperature)


def concrete_binary_kl_mc_sample(y, prior_log_odds, prior_temperature,
                                 posterior_log_odds, posterior_temperature):
    """concrete.py:30-43."""
    f = y.dtype.type
    pt, qt = f(prior_temperature), f(posterior_temperature)
    plo = f(prior_log_odds) if np.isscalar(prior_log_odds) or np.ndim(prior_log_odds) == 0 else prior_log_odds
    y_p = y * pt
    log_prior = np.log(pt + f(EPS)) - y_p + plo - \
        f(2.0) * np.log(f(1.0) + np.exp(-y_p + plo) + f(EPS))
    y_q = y * qt
    log_post = np.log(qt + f(EPS)) - y_q + posterior_log_odds - \
        f(2.0) * np.log(f(1.0) + np.exp(-y_q + posterior_log_odds) + f(EPS))
    return log_post - log_prior


# --------------------------------------------------------------------------- #
# VAE -- air/vae.py
# --------------------------------------------------------------------------- #

def vae(inputs, params, hp, eps_z, eps_x):
    """vae.py:5-43.  Returns (reconstruction, rec_mean, rec_log_variance,
    rec_mean) -- the 4th value is the MEAN, not the sample (vae.py:43)."""
    f = inputs.dtype.type
    h = inputs
    for i in range(len(hp["vae_recognition_units"])):
        h = fully_connected(h, params["vae/recognition_%d/weights" % (i + 1)],
                            params["vae/recognition_%d/biases" % (i + 1)], "softplus")
    rec_mean = fully_connected(h, params["vae/rec_mean/weights"], params["vae/rec_mean/biases"])
    rec_lv = fully_connected(h, params["vae/rec_log_variance/weights"], params["vae/rec_log_variance/biases"])
    sample = rec_mean + eps_z * np.sqrt(np.exp(rec_lv))           # vae.py:22-24
    h = sample
    for i in range(len(hp["vae_generative_units"])):
        h = fully_connected(h, params["vae/generative_%d/weights" % (i + 1)],
                            params["vae/generative_%d/biases" % (i + 1)], "softplus")
    gen_mean = fully_connected(h, params["vae/gen_mean/weights"], params["vae/gen_mean/biases"])
    gen_sample = gen_mean + eps_x * f(hp["vae_likelihood_std"])     # vae.py:36-38
    recon = sigmoid(gen_sample)                                     # vae.py:39-41
    return recon, rec_mean, rec_lv, rec_mean


# --------------------------------------------------------------------------- #
# LSTM -- tf.contrib.rnn.BasicLSTMCell (TF 1.3), call site air_model.py:286,539
# --------------------------------------------------------------------------- #

def lstm_cell(x, c, h, kernel, bias, forget_bias=1.0):
    """gates = [x, h] @ kernel + bias; i, j, f, o = split(gates, 4, axis=1);
    c' = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j);
    h' = tanh(c') * sigmoid(o).  Gate order and forget_bias = 1.0 are read from
    air-model.meta nodes air/rnn/while/rnn/{split, add/y, Sigmoid*, Tanh*}."""
    fb = x.dtype.type(forget_bias)
    g = np.concatenate([x, h], axis=1) @ kernel + bias
    i, j, f, o = np.split(g, 4, axis=1)
    new_c = c * sigmoid(f + fb) + sigmoid(i) * np.tanh(j)
    new_h = np.tanh(new_c) * sigmoid(o)
    return new_c, new_h


# --------------------------------------------------------------------------- #
# the while-loop body + loss -- air/air_model.py:269-611
# --------------------------------------------------------------------------- #

def _gauss_kl(prior_log_var, log_var, var, prior_var, mean, prior_mean):
    """air_model.py:443-447 / 462-466 / 481-485."""
    f = mean.dtype.type
    return f(0.5) * np.sum(
        f(prior_log_var) - log_var - f(1.0) + var / f(prior_var) +
        np.square(mean - f(prior_mean)) / f(prior_var), axis=1)


def air_forward(params, images, targets, noise, hp, train=True,
                z_pres_prior_log_odds=None, early_exit=False):
    """AIRModel._create_model, air_model.py:269-611, with injected noise.

    early_exit=True reproduces cond (:271-275): the stacked outputs then have
    leading time dimension T' <= max_steps.  early_exit=False runs a fixed N
    steps; loss / reconstruction / counts are identical (finished items are
    masked, :411-415, 433-439, 451-455, 470-474, 489-493)."""
    dtype = images.dtype
    f = dtype.type
    B = images.shape[0]
    N = hp["max_steps"]
    C, w = hp["canvas_size"], hp["windows_size"]
    thr = f(hp["stopping_threshold"])
    temp = hp["z_pres_temperature"]
    prior_lo = f(hp["z_pres_prior_log_odds"] if z_pres_prior_log_odds is None else z_pres_prior_log_odds)

    # priors' log-variances, air_model.py:72-74 (tf.log of python floats, fp32)
    scale_plv = np.log(f(hp["scale_prior_variance"]))
    shift_plv = np.log(f(hp["shift_prior_variance"]))
    vae_plv = np.log(f(hp["vae_prior_variance"]))

    S = np.zeros(B, dtype)                     # stopping_sum           :550
    c = np.zeros((B, hp["rnn_units"]), dtype)  # LSTM zero_state        :540
    h = np.zeros((B, hp["rnn_units"]), dtype)
    R = np.zeros_like(images)                  # running_recon          :552
    L = np.zeros(B, dtype)                     # running_loss           :553
    digits = np.zeros(B, np.int32)             # running_digits         :554
    canvas = images.reshape(B, C, C)

    keys = ("scales", "shifts", "z_pres_probs", "z_pres_kls", "scale_kls", "shift_kls",
            "vae_kls", "st_back", "windows", "latents", "z_pres", "z_pres_pre_sigmoid",
            "window_in", "window_recon")
    ta = {k: [] for k in keys}

    step = 0
    while step < N and (not early_exit or np.any(S < thr)):            # cond :271-275
        t = step
        c, h = lstm_cell(images, c, h, params["rnn/kernel"], params["rnn/bias"])  # :286
        outputs = h

        def head(name, act_hidden="relu"):
            hid = fully_connected(outputs, params[name + "/hidden/weights"],
                                  params[name + "/hidden/biases"], act_hidden)
            return fully_connected(hid, params[name + "/output/weights"], params[name + "/output/biases"])

        # scale :288-303
        scale_mean = head("scale/mean")
        scale_lv = head("scale/log_variance")
        scale_var = np.exp(scale_lv)
        scale = sigmoid(scale_mean + noise["eps_scale"][t] * np.sqrt(scale_var))
        s = scale[:, 0]
        # shift :305-320
        shift_mean = head("shift/mean")
        shift_lv = head("shift/log_variance")
        shift_var = np.exp(shift_lv)
        shift = np.tanh(shift_mean + noise["eps_shift"][t] * np.sqrt(shift_var))
        x, y = shift[:, 0], shift[:, 1]

        # st_forward :322-333
        zeros = np.zeros_like(s)
        theta = np.stack([np.stack([s, zeros, x], axis=1), np.stack([zeros, s, y], axis=1)], axis=1)
        window = transformer(canvas, theta, (w, w))

        # vae :335-349
        vae_recon, vae_mean, vae_lv, vae_latent = vae(
            window.reshape(B, w * w), params, hp, noise["eps_z"][t], noise["eps_x"][t])

        # st_backward :351-366
        theta_recon = np.stack([np.stack([f(1.0) / s, zeros, -x / s], axis=1),
                                np.stack([zeros, f(1.0) / s, -y / s], axis=1)], axis=1)
        window_recon = transformer(vae_recon.reshape(B, w, w), theta_recon, (C, C))

        # z_pres :368-396
        z_lo = head("z_pres/log_odds")[:, 0]
        z_pre = concrete_binary_pre_sigmoid_sample(z_lo, temp, noise["u"][t])
        z_pres = sigmoid(z_pre)
        if not train:
            z_pres = np.round(z_pres)                    # tf.round: half-to-even, :389-390
        z_prob = sigmoid(z_lo)

        # loss/z_pres_kl :398-418 -- masked by the PREVIOUS stopping sum
        z_kl = concrete_binary_kl_mc_sample(z_pre, prior_lo, temp, z_lo, temp)
        L = L + np.where(S < thr, z_kl, np.zeros_like(L))

        S = S + (f(1.0) - z_pres)                                           # :424
        active = S < thr
        digits = digits + active.astype(np.int32)                           # :427

        # canvas :429-439
        R = R + np.where(active[:, None], z_pres[:, None] * window_recon.reshape(B, C * C), np.zeros_like(R))

        # KLs :441-496 -- masked by the UPDATED stopping sum
        scale_kl = _gauss_kl(scale_plv, scale_lv, scale_var, hp["scale_prior_variance"],
                             scale_mean, hp["scale_prior_mean"])
        L = L + np.where(active, scale_kl, np.zeros_like(L))
        shift_kl = _gauss_kl(shift_plv, shift_lv, shift_var, hp["shift_prior_variance"],
                             shift_mean, hp["shift_prior_mean"])
        L = L + np.where(active, shift_kl, np.zeros_like(L))
        vae_kl = _gauss_kl(vae_plv, vae_lv, np.exp(vae_lv), hp["vae_prior_variance"],
                           vae_mean, hp["vae_prior_mean"])
        L = L + np.where(active, vae_kl, np.zeros_like(L))

        for k, v in (("scales", scale), ("shifts", shift), ("z_pres_probs", z_prob),
                     ("z_pres_kls", z_kl), ("scale_kls", scale_kl), ("shift_kls", shift_kl),
                     ("vae_kls", vae_kl), ("st_back", theta_recon), ("windows", vae_recon),
                     ("latents", vae_latent), ("z_pres", z_pres), ("z_pres_pre_sigmoid", z_pre),
                     ("window_in", window.reshape(B, w * w)),
                     ("window_recon", window_recon.reshape(B, C * C))):
            ta[k].append(v)
        step += 1

    out = {}
    # stack + transpose :568-578
    out["rec_scales"] = np.transpose(np.stack(ta["scales"]), (1, 0, 2))
    out["rec_shifts"] = np.transpose(np.stack(ta["shifts"]), (1, 0, 2))
    out["rec_st_back"] = np.transpose(np.stack(ta["st_back"]), (1, 0, 2, 3))
    out["rec_windows"] = np.transpose(np.stack(ta["windows"]), (1, 0, 2))
    out["rec_latents"] = np.transpose(np.stack(ta["latents"]), (1, 0, 2))
    for k in ("z_pres_probs", "z_pres_kls", "scale_kls", "shift_kls", "vae_kls"):
        out[k] = np.transpose(np.stack(ta[k]))
    # diagnostics that are not reference attributes (used by kernel-level parity tests)
    out["_z_pres"] = np.transpose(np.stack(ta["z_pres"]))
    out["_z_pres_pre_sigmoid"] = np.transpose(np.stack(ta["z_pres_pre_sigmoid"]))
    out["_window_in"] = np.transpose(np.stack(ta["window_in"]), (1, 0, 2))
    out["_window_recon"] = np.transpose(np.stack(ta["window_recon"]), (1, 0, 2))
    out["_running_recon"] = R
    out["_running_loss"] = L

    # loss/reconstruction :580-593
    recon = np.maximum(np.minimum(R, f(1.0)), f(0.0))
    rec_loss = -np.sum(images * np.log(recon + f(EPS)) +
                       (f(1.0) - images) * np.log(f(1.0) - recon + f(EPS)), axis=1)
    loss_vec = L + rec_loss
    out["reconstruction"] = recon
    out["reconstruction_loss"] = rec_loss
    out["rec_num_digits"] = digits
    out["loss_per_item"] = loss_vec
    out["loss"] = np.mean(loss_vec)                                        # :610
    out["accuracy"] = np.mean((targets == digits).astype(dtype))           # :597-611
    out["steps_executed"] = step
    return out


# --------------------------------------------------------------------------- #
# optimizer -- air_model.py:651-694 (tf.clip_by_global_norm + TF1.3 ApplyAdam)
# --------------------------------------------------------------------------- #

def clip_by_global_norm(grads, clip_norm):
    """tf.clip_by_global_norm: global_norm = sqrt(sum_i 2*l2_loss(g_i));
    scale = clip_norm * min(1/global_norm, 1/clip_norm)."""
    any_g = next(iter(grads.values()))
    f = any_g.dtype.type
    half_sq = [np.sum(np.square(g)) / f(2.0) for g in grads.values()]
    gn = np.sqrt(f(2.0) * np.sum(np.array(half_sq, dtype=any_g.dtype)))
    scale = f(clip_norm) * np.minimum(f(1.0) / gn, f(1.0) / f(clip_norm))
    return OrderedDict((k, g * scale) for k, g in grads.items()), gn


def adam_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """TF 1.3 training_ops ApplyAdam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= lr_t*m/(sqrt(v)+eps).
    epsilon sits OUTSIDE the bias correction (differs from torch.optim.Adam).
    t is the 1-based step count (beta power accumulators after t updates)."""
    any_p = next(iter(params.values()))
    f = any_p.dtype.type
    b1, b2 = f(beta1), f(beta2)
    lr_t = f(lr) * np.sqrt(f(1.0) - np.power(b2, f(t))) / (f(1.0) - np.power(b1, f(t)))
    for k in params:
        g = grads[k]
        m[k] = m[k] + (g - m[k]) * (f(1.0) - b1)
        v[k] = v[k] + (np.square(g) - v[k]) * (f(1.0) - b2)
        params[k] = params[k] - (m[k] * lr_t) / (np.sqrt(v[k]) + f(epsilon))
    return params, m, v
