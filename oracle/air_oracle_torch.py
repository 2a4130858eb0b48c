"""PyTorch-CPU autograd twin of ``air_oracle.py`` -- TEST INFRASTRUCTURE ONLY.

Same restatement of the reference's op sequence (air/air_model.py:269-611,
air/transformer.py:56-171, air/vae.py:5-43, air/concrete.py:20-43) but in
torch so that autograd supplies the reference's gradients
(``optimizer.compute_gradients(self.loss)``, air_model.py:655) and the clipped
TF-style Adam update (:673-694).  It is deliberately UN-fused and recomputes
``concat([x, h]) @ kernel`` every step exactly as the reference graph does, so
it doubles as ``bench.py``'s ``cpu_baseline`` (kind "port").

PARITY UNPINNED BY REFERENCE TESTS -- see air_oracle.py header.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

EPS = 10e-10


class _Bf16MatMul(torch.autograd.Function):
    """Emulates the device's bf16-operand / fp32-accumulate GEMMs: forward AND both
    gradient GEMMs round their two operands to bf16 (air_gemm precision = 1)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _r(a) @ _r(b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return _r(g) @ _r(b).t(), _r(a).t() @ _r(g)


class _HeadOutMatMul(torch.autograd.Function):
    """The 7 head output units on the device's bf16 path: forward and data gradient are exact fp32
    (computed inside attend_fwd / attend_bwd, not by a GEMM); only their weight gradient goes through the
    grouped bf16 weight-gradient launch (both operands rounded)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return a @ b

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return g @ b.t(), _r(a).t() @ _r(g)


def _r(x):
    return x.to(torch.bfloat16).to(x.dtype)


MATMUL_MODE = "exact"      # "bf16": emulate the device's bf16 GEMM path (tests only)


def _mm(a, b):
    return _Bf16MatMul.apply(a, b) if MATMUL_MODE == "bf16" else a @ b


def _fc(x, W, b, act=None, head_out=False):
    if head_out and MATMUL_MODE == "bf16":
        y = _HeadOutMatMul.apply(x, W) + b
    else:
        y = _mm(x, W) + b
    if act == "relu":
        return torch.relu(y)
    if act == "softplus":
        return torch.nn.functional.softplus(y, threshold=13.9)
    return y


_CONSTS = {}


def _const(value, dtype):
    """a 0-d tensor of `value` on the default device, made once: a host-to-device copy per use would be the one thing in
    this op sequence a hipGraph capture (tools/twin_train_gpu.py) cannot record"""
    key = (float(value), dtype, str(torch.get_default_device()) if hasattr(torch, "get_default_device") else "")
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(float(value), dtype=dtype)
    return _CONSTS[key]


def _linspace(n, dtype):
    if n == 1:
        return torch.tensor([-1.0], dtype=dtype)
    step = _const(2.0, dtype) / _const(float(n - 1), dtype)
    return _const(-1.0, dtype) + step * torch.arange(n, dtype=dtype)


def transformer(U, theta, out_size):
    """transformer.py:56-171 -- gradient flows through the weights (x, y) and
    the gathered taps; floor/clip are integer ops."""
    dtype = U.dtype
    B, Hi, Wi = U.shape
    Ho, Wo = out_size
    x_t = _linspace(Wo, dtype).repeat(Ho)
    y_t = _linspace(Ho, dtype).repeat_interleave(Wo)
    x_s = (theta[:, 0, 0:1] * x_t[None] + theta[:, 0, 1:2] * y_t[None]) + theta[:, 0, 2:3]
    y_s = (theta[:, 1, 0:1] * x_t[None] + theta[:, 1, 1:2] * y_t[None]) + theta[:, 1, 2:3]
    cx = _const(float(Wi), dtype) - _const(1.001, dtype)
    cy = _const(float(Hi), dtype) - _const(1.001, dtype)
    x = (x_s + 1.0) * cx / 2.0
    y = (y_s + 1.0) * cy / 2.0
    x0 = torch.floor(x.detach()).to(torch.int64)
    y0 = torch.floor(y.detach()).to(torch.int64)
    x1 = (x0 + 1).clamp(0, Wi - 1)
    y1 = (y0 + 1).clamp(0, Hi - 1)
    x0 = x0.clamp(0, Wi - 1)
    y0 = y0.clamp(0, Hi - 1)
    flat = U.reshape(B, Hi * Wi)
    Ia = torch.gather(flat, 1, y0 * Wi + x0)
    Ib = torch.gather(flat, 1, y1 * Wi + x0)
    Ic = torch.gather(flat, 1, y0 * Wi + x1)
    Id = torch.gather(flat, 1, y1 * Wi + x1)
    x0f, x1f, y0f, y1f = x0.to(dtype), x1.to(dtype), y0.to(dtype), y1.to(dtype)
    wa = (x1f - x) * (y1f - y)
    wb = (x1f - x) * (y - y0f)
    wc = (x - x0f) * (y1f - y)
    wd = (x - x0f) * (y - y0f)
    out = ((wa * Ia + wb * Ib) + wc * Ic) + wd * Id
    return out.reshape(B, Ho, Wo)


def _gauss_kl(plv, lv, var, pvar, mean, pmean):
    return 0.5 * torch.sum(plv - lv - 1.0 + var / pvar + (mean - pmean) ** 2 / pvar, dim=1)


def _concrete_kl(y, prior_lo, pt, post_lo, qt):
    y_p = y * pt
    log_prior = math.log(pt + EPS) - y_p + prior_lo - 2.0 * torch.log(1.0 + torch.exp(-y_p + prior_lo) + EPS)
    y_q = y * qt
    log_post = math.log(qt + EPS) - y_q + post_lo - 2.0 * torch.log(1.0 + torch.exp(-y_q + post_lo) + EPS)
    return log_post - log_prior


def air_forward(params, images, targets, noise, hp, train=True, z_pres_prior_log_odds=None):
    """Fixed-N restatement of air_model.py:269-611 (see air_oracle.air_forward)."""
    dtype = images.dtype
    B = images.shape[0]
    N = hp["max_steps"]
    C, w = hp["canvas_size"], hp["windows_size"]
    R_units = hp["rnn_units"]
    thr = hp["stopping_threshold"]
    temp = hp["z_pres_temperature"]
    prior_lo = hp["z_pres_prior_log_odds"] if z_pres_prior_log_odds is None else (
        z_pres_prior_log_odds if torch.is_tensor(z_pres_prior_log_odds) else float(z_pres_prior_log_odds))
    scale_plv = math.log(hp["scale_prior_variance"])
    shift_plv = math.log(hp["shift_prior_variance"])
    vae_plv = math.log(hp["vae_prior_variance"])

    S = torch.zeros(B, dtype=dtype)
    c = torch.zeros(B, R_units, dtype=dtype)
    h = torch.zeros(B, R_units, dtype=dtype)
    R = torch.zeros_like(images)
    L = torch.zeros(B, dtype=dtype)
    digits = torch.zeros(B, dtype=torch.int32)
    canvas = images.reshape(B, C, C)
    stacks = {k: [] for k in ("scales", "shifts", "z_pres_probs", "z_pres_kls", "scale_kls",
                              "shift_kls", "vae_kls", "st_back", "windows", "latents")}

    def head(outputs, name):
        hid = _fc(outputs, params[name + "/hidden/weights"], params[name + "/hidden/biases"], "relu")
        return _fc(hid, params[name + "/output/weights"], params[name + "/output/biases"], head_out=True)

    for t in range(N):
        g = _mm(torch.cat([images, h], dim=1), params["rnn/kernel"]) + params["rnn/bias"]
        gi, gj, gf, go = torch.split(g, R_units, dim=1)
        c = c * torch.sigmoid(gf + 1.0) + torch.sigmoid(gi) * torch.tanh(gj)
        h = torch.tanh(c) * torch.sigmoid(go)
        outputs = h

        scale_mean = head(outputs, "scale/mean")
        scale_lv = head(outputs, "scale/log_variance")
        scale_var = torch.exp(scale_lv)
        scale = torch.sigmoid(scale_mean + noise["eps_scale"][t] * torch.sqrt(scale_var))
        s = scale[:, 0]
        shift_mean = head(outputs, "shift/mean")
        shift_lv = head(outputs, "shift/log_variance")
        shift_var = torch.exp(shift_lv)
        shift = torch.tanh(shift_mean + noise["eps_shift"][t] * torch.sqrt(shift_var))
        x, y = shift[:, 0], shift[:, 1]

        zeros = torch.zeros_like(s)
        theta = torch.stack([torch.stack([s, zeros, x], 1), torch.stack([zeros, s, y], 1)], 1)
        window = transformer(canvas, theta, (w, w)).reshape(B, w * w)

        hh = window
        for i in range(len(hp["vae_recognition_units"])):
            hh = _fc(hh, params["vae/recognition_%d/weights" % (i + 1)],
                     params["vae/recognition_%d/biases" % (i + 1)], "softplus")
        vae_mean = _fc(hh, params["vae/rec_mean/weights"], params["vae/rec_mean/biases"])
        vae_lv = _fc(hh, params["vae/rec_log_variance/weights"], params["vae/rec_log_variance/biases"])
        hh = vae_mean + noise["eps_z"][t] * torch.sqrt(torch.exp(vae_lv))
        for i in range(len(hp["vae_generative_units"])):
            hh = _fc(hh, params["vae/generative_%d/weights" % (i + 1)],
                     params["vae/generative_%d/biases" % (i + 1)], "softplus")
        gen_mean = _fc(hh, params["vae/gen_mean/weights"], params["vae/gen_mean/biases"])
        vae_recon = torch.sigmoid(gen_mean + noise["eps_x"][t] * hp["vae_likelihood_std"])

        theta_recon = torch.stack([torch.stack([1.0 / s, zeros, -x / s], 1),
                                   torch.stack([zeros, 1.0 / s, -y / s], 1)], 1)
        window_recon = transformer(vae_recon.reshape(B, w, w), theta_recon, (C, C)).reshape(B, C * C)

        z_lo = head(outputs, "z_pres/log_odds")[:, 0]
        u = noise["u"][t]
        z_pre = (z_lo + (torch.log(u + EPS) - torch.log(1.0 - u + EPS))) / temp
        z_pres = torch.sigmoid(z_pre)
        if not train:
            z_pres = torch.round(z_pres)
        z_kl = _concrete_kl(z_pre, prior_lo, temp, z_lo, temp)
        L = L + torch.where(S < thr, z_kl, torch.zeros_like(L))
        S = S + (1.0 - z_pres).detach()  # S only feeds comparisons: no gradient path
        active = S < thr
        digits = digits + active.to(torch.int32)
        R = R + torch.where(active[:, None], z_pres[:, None] * window_recon, torch.zeros_like(R))
        scale_kl = _gauss_kl(scale_plv, scale_lv, scale_var, hp["scale_prior_variance"],
                             scale_mean, hp["scale_prior_mean"])
        L = L + torch.where(active, scale_kl, torch.zeros_like(L))
        shift_kl = _gauss_kl(shift_plv, shift_lv, shift_var, hp["shift_prior_variance"],
                             shift_mean, hp["shift_prior_mean"])
        L = L + torch.where(active, shift_kl, torch.zeros_like(L))
        vae_kl = _gauss_kl(vae_plv, vae_lv, torch.exp(vae_lv), hp["vae_prior_variance"],
                           vae_mean, hp["vae_prior_mean"])
        L = L + torch.where(active, vae_kl, torch.zeros_like(L))
        for k, v in (("scales", scale), ("shifts", shift), ("z_pres_probs", torch.sigmoid(z_lo)),
                     ("z_pres_kls", z_kl), ("scale_kls", scale_kl), ("shift_kls", shift_kl),
                     ("vae_kls", vae_kl), ("st_back", theta_recon), ("windows", vae_recon),
                     ("latents", vae_mean)):
            stacks[k].append(v)

    recon = torch.clamp(R, 0.0, 1.0)     # Minimum/Maximum grads pass at ties == clamp
    rec_loss = -torch.sum(images * torch.log(recon + EPS) +
                          (1.0 - images) * torch.log(1.0 - recon + EPS), dim=1)
    loss_vec = L + rec_loss
    out = dict(loss=loss_vec.mean(), loss_per_item=loss_vec, reconstruction=recon,
               reconstruction_loss=rec_loss, rec_num_digits=digits,
               accuracy=(targets.to(torch.int32) == digits).to(dtype).mean())
    out["rec_scales"] = torch.stack(stacks["scales"]).transpose(0, 1)
    out["rec_shifts"] = torch.stack(stacks["shifts"]).transpose(0, 1)
    out["rec_st_back"] = torch.stack(stacks["st_back"]).transpose(0, 1)
    out["rec_windows"] = torch.stack(stacks["windows"]).transpose(0, 1)
    out["rec_latents"] = torch.stack(stacks["latents"]).transpose(0, 1)
    for k in ("z_pres_probs", "z_pres_kls", "scale_kls", "shift_kls", "vae_kls"):
        out[k] = torch.stack(stacks[k]).t()
    return out


def to_torch(d, dtype=torch.float32, requires_grad=False):
    out = OrderedDict()
    for k, v in d.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def loss_and_grads(params, images, targets, noise, hp, z_pres_prior_log_odds=None):
    """optimizer.compute_gradients(self.loss), air_model.py:655."""
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    out = air_forward(params, images, targets, noise, hp, True, z_pres_prior_log_odds)
    grads = torch.autograd.grad(out["loss"], list(params.values()), allow_unused=False)
    return out, OrderedDict(zip(params.keys(), grads))


def clip_and_adam(params, grads, m, v, t, hp, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """tf.clip_by_global_norm (air_model.py:673) then TF1.3 ApplyAdam (:692)."""
    c = hp["gradient_clipping_norm"]
    gn = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads.values()))
    scale = c * torch.minimum(1.0 / gn, torch.tensor(1.0 / c, dtype=gn.dtype))
    lr_t = hp["learning_rate"] * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k] * scale
            m[k] += (g - m[k]) * (1.0 - beta1)
            v[k] += (g * g - v[k]) * (1.0 - beta2)
            p -= (m[k] * lr_t) / (torch.sqrt(v[k]) + epsilon)
    return gn


class CpuTrainer:
    """Un-fused CPU train step used as bench.py's cpu_baseline ("port")."""

    def __init__(self, params_np, hp, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.hp = hp
        self.params = to_torch(params_np, requires_grad=True)
        self.m = OrderedDict((k, torch.zeros_like(p)) for k, p in self.params.items())
        self.v = OrderedDict((k, torch.zeros_like(p)) for k, p in self.params.items())
        self.t = 0
        self.gen = torch.Generator().manual_seed(0)

    def draw_noise(self, B):
        hp = self.hp
        N, Z, d = hp["max_steps"], hp["vae_latent_dimensions"], hp["windows_size"] ** 2
        g = self.gen
        return dict(eps_scale=torch.randn(N, B, 1, generator=g), eps_shift=torch.randn(N, B, 2, generator=g),
                    eps_z=torch.randn(N, B, Z, generator=g), eps_x=torch.randn(N, B, d, generator=g),
                    u=torch.rand(N, B, generator=g))

    def step(self, images, targets, prior_lo=None):
        noise = self.draw_noise(images.shape[0])
        out, grads = loss_and_grads(self.params, images, targets, noise, self.hp, prior_lo)
        self.t += 1
        clip_and_adam(self.params, grads, self.m, self.v, self.t, self.hp)
        return float(out["loss"]), float(out["accuracy"])
