"""Diagnostic (not the product path): trains the torch-autograd twin of the reference
(oracle/air_oracle_torch.py: un-fused fp32 ops, autograd's own residue-carrying gradients) with
torch-ROCm kernels on the same data, annealing and optimizer as training.py -- an independent
implementation of "the reference's fp32 autodiff" to compare success rates with.
  python tools/twin_train_gpu.py <seed> <iterations> [<backgrounds.npz>:<key> [<max intensity>]]   (clutter: BASELINE configs[4])"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from oracle import air_oracle as ao, air_oracle_torch as at
from multi_mnist import generate_dataset, shift_zero_digits_images

seed, iters = int(sys.argv[1]), int(sys.argv[2])
torch.set_default_device("cuda")
hp = dict(ao.TRAINING_HP)
bg = None
if len(sys.argv) > 3:                                      # clutter background, already scaled (tests/golden/backgrounds.npz)
    f_, key_ = sys.argv[3].rsplit(":", 1)
    bg = np.load(f_)[key_].astype(np.float32)
    if len(sys.argv) > 4 and bg.max() > 0:                 # rescaled to a maximum intensity, as training.py --bg-max-intensity does
        bg = bg / bg.max() * min(float(sys.argv[4]), 1.0)
ds = generate_dataset(bg=bg)
te_im, te_dg = shift_zero_digits_images(ds["test_images"], ds["test_digits"])
tr_im, tr_dg = torch.tensor(ds["train_images"]), torch.tensor(ds["train_digits"].astype(np.int32))
te_im, te_dg = torch.tensor(np.ascontiguousarray(te_im)), torch.tensor(np.ascontiguousarray(te_dg).astype(np.int32))
params = {k: torch.tensor(v, requires_grad=True) for k, v in ao.init_params(hp, seed).items()}
m = {k: torch.zeros_like(p) for k, p in params.items()}
v = {k: torch.zeros_like(p) for k, p in params.items()}
g = torch.Generator(device="cuda").manual_seed(seed)
N, Z, d, B = hp["max_steps"], hp["vae_latent_dimensions"], hp["windows_size"] ** 2, 64
sched = ao.TRAINING_ANNEALING["z_pres_prior_log_odds"]

def noise(b):
    return dict(eps_scale=torch.randn(N, b, 1, generator=g), eps_shift=torch.randn(N, b, 2, generator=g),
                eps_z=torch.randn(N, b, Z, generator=g), eps_x=torch.randn(N, b, d, generator=g),
                u=torch.rand(N, b, generator=g))

def evaluate(lo):
    with torch.no_grad():
        o = at.air_forward(params, te_im, te_dg, noise(len(te_im)), hp, False, lo)
    dig = o["rec_num_digits"]
    acc = [float((dig[te_dg == k] == k).float().mean()) for k in range(3)]
    return float((dig == te_dg).float().mean()), acc

t0 = time.time()
perm = torch.randperm(len(tr_im), generator=g)
ptr = 0
for it in range(iters):
    lo = float(ao.annealed_value(sched, it))
    if it % 5000 == 0:
        a, acc = evaluate(lo)
        print(json.dumps({"seed": seed, "step": it, "accuracy": round(a, 3), "acc012": [round(x, 2) for x in acc],
                          "wall_s": round(time.time() - t0, 1)}), flush=True)
    if ptr + B > len(tr_im):
        perm, ptr = torch.randperm(len(tr_im), generator=g), 0
    idx = perm[ptr:ptr + B]; ptr += B
    out, grads = at.loss_and_grads(params, tr_im[idx], tr_dg[idx], noise(B), hp, lo)
    t = it + 1
    c = hp["gradient_clipping_norm"]
    gn = torch.sqrt(sum((x.detach() ** 2).sum() for x in grads.values()))
    scale = c * torch.clamp(1.0 / gn, max=1.0 / c)
    lr_t = hp["learning_rate"] * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
    with torch.no_grad():
        for k, p in params.items():
            gk = grads[k] * scale
            m[k] += (gk - m[k]) * 0.1
            v[k] += (gk * gk - v[k]) * 0.001
            p -= (m[k] * lr_t) / (torch.sqrt(v[k]) + 1e-8)
a, acc = evaluate(float(ao.annealed_value(sched, iters)))
print(json.dumps({"seed": seed, "step": iters, "accuracy": round(a, 3), "acc012": [round(x, 2) for x in acc],
                  "wall_s": round(time.time() - t0, 1), "final": True}), flush=True)
