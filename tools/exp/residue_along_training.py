"""Along a real training run (training.py's model and data, backward="reference"): every K iterations the write backward is
evaluated a second time on the SAME buffers in the carried / blocked order, and the size of d_gen_pre at the four corner
pixels, the border pixels and the interior is recorded for all three -- does the carried order keep the residue's size
beyond initialisation?   python tools/exp/residue_along_training.py [iterations] [every]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
from air import _hip as H  # noqa: E402
from air.air_model import AIRModel  # noqa: E402
from multi_mnist import generate_dataset  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
ds = generate_dataset()
tr = torch.tensor(ds["train_images"], device=dev)
tg = torch.tensor(ds["train_digits"].astype(np.int32), device=dev)
B, Cc, w, N = 64, 50, 28, 3
x, t = torch.zeros(B, Cc * Cc, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
m = AIRModel(x, t, max_steps=3, max_digits=2, rnn_units=256, canvas_size=50, windows_size=28, vae_latent_dimensions=50,
             vae_recognition_units=(512, 256), vae_generative_units=(256, 512), scale_prior_mean=-1.0, scale_prior_variance=0.05,
             shift_prior_mean=0.0, shift_prior_variance=1.0, vae_prior_mean=0.0, vae_prior_variance=1.0, vae_likelihood_std=0.3,
             scale_hidden_units=64, shift_hidden_units=64, z_pres_hidden_units=64, z_pres_prior_log_odds=-0.01,
             z_pres_temperature=1.0, stopping_threshold=0.99, learning_rate=1e-4, gradient_clipping_norm=1.0, cnn=False,
             train=True, scope="air", gemm_precision="bf16", backward="reference", seed=0,
             annealing_schedules={"z_pres_prior_log_odds": {"init": 10000.0, "min": 1e-9, "factor": 0.1, "iters": 3000,
                                                            "staircase": False, "log": True}})
p = lambda v: C.c_void_p(v.data_ptr())  # noqa: E731
corn = [0, w - 1, w * (w - 1), w * w - 1]
border = [i for i in range(w * w) if (i // w in (0, w - 1) or i % w in (0, w - 1)) and i not in corn]
inner = [i for i in range(w * w) if i not in corn and i not in border]
g = torch.Generator(device=dev).manual_seed(0)
rms = lambda a, idx: float(a[:, :, idx].double().pow(2).mean().sqrt())  # noqa: E731
for it in range(iters + 1):
    idx = torch.randint(0, tr.shape[0], (B,), device=dev, generator=g)
    x.copy_(tr[idx]); t.copy_(tg[idx])
    m.training(eager=True)
    if it % every == 0:
        torch.cuda.synchronize()
        row = {"step": it, "gnorm": float(m.store.gnorm[0])}
        ref = m.d_genpre.clone()
        for name, lit in (("reference", 2), ("carried", 4), ("blocked", 3)):
            dgen, dsx = torch.zeros_like(m.d_genpre), torch.zeros_like(m.d_sxyw)
            wb = H.WriteBwd(p(m.d_recon), p(m.vrec), p(m.att), p(dgen), p(dsx), B, N, Cc, w, lit, None, None, None, None)
            H.check(H.lib().air_write_bwd(C.byref(wb), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            if lit == 2:
                assert torch.equal(dgen, ref)
            row[name] = {"corner": rms(dgen, corn), "border": rms(dgen, border), "interior": rms(dgen, inner)}
        print(json.dumps(row), flush=True)
