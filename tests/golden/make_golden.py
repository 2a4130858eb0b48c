"""Generates tests/golden/air_b4.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

The reference cannot be executed (TensorFlow 1.3 is not installable here), so
these vectors are a snapshot of the oracle restatement -- they pin the oracle
against regressions; they are NOT outputs of the reference itself (parity
unpinned, see oracle/air_oracle.py header).  Inputs are regenerated from seeds
(np.random.RandomState is platform-stable), outputs are stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import air_oracle as ao  # noqa: E402
from oracle import air_oracle_torch as at  # noqa: E402
from oracle.synth import blob_canvases  # noqa: E402
import torch  # noqa: E402

KEYS = ("loss", "accuracy", "loss_per_item", "reconstruction", "reconstruction_loss", "rec_num_digits",
        "rec_scales", "rec_shifts", "rec_st_back", "rec_windows", "rec_latents",
        "z_pres_probs", "z_pres_kls", "scale_kls", "shift_kls", "vae_kls")


def main():
    hp = dict(ao.TRAINING_HP)
    B = 4
    images, targets = blob_canvases(B, hp["canvas_size"], hp["max_digits"], seed=3)
    params = ao.init_params(hp, seed=0)
    noise = ao.make_noise(hp, B, seed=1)
    out = {}
    for tag, train, lo in (("train_lo9", True, ao.annealed_value(ao.TRAINING_ANNEALING["z_pres_prior_log_odds"], 0)),
                           ("train_lom2", True, np.float32(-2.0)),
                           ("test_lom2", False, np.float32(-2.0))):
        o = ao.air_forward(params, images, targets, noise, hp, train, lo)
        for k in KEYS:
            out["%s/%s" % (tag, k)] = np.asarray(o[k])
    # gradients + one clipped Adam step from the autograd twin (train, prior log-odds -2)
    pt = at.to_torch(params, requires_grad=True)
    o, grads = at.loss_and_grads(pt, torch.tensor(images), torch.tensor(targets), at.to_torch(noise), hp, -2.0)
    out["grad/loss"] = np.float32(o["loss"].item())
    for k, g in grads.items():
        out["grad_norm/" + k] = np.float32(g.norm().item())
        out["grad_sum/" + k] = np.float32(g.double().sum().item())
    out["grad/rnn_bias"] = grads["rnn/bias"].numpy()
    out["grad/gen_mean_biases"] = grads["vae/gen_mean/biases"].numpy()
    out["grad/scale_mean_out_w"] = grads["scale/mean/output/weights"].numpy()
    m = {k: torch.zeros_like(p) for k, p in pt.items()}
    v = {k: torch.zeros_like(p) for k, p in pt.items()}
    gn = at.clip_and_adam(pt, grads, m, v, 1, hp)
    out["adam/global_norm"] = np.float32(gn.item())
    out["adam/rnn_bias_after"] = pt["rnn/bias"].detach().numpy()
    out["adam/gen_mean_biases_after"] = pt["vae/gen_mean/biases"].detach().numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "air_b4.npz"), **out)
    print("wrote", len(out), "arrays; loss(train, lo=-2) =", out["train_lom2/loss"])


if __name__ == "__main__":
    main()
