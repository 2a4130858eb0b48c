import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from air import air_model as am
from oracle import air_oracle as ao, air_oracle_torch as at
import multi_mnist as mm
HP = dict(ao.TRAINING_HP)
ds = mm.generate_dataset(2, 300, 10)
B = 64
images = ds["train_images"][:B]; targets = ds["train_digits"][:B]
params = ao.init_params(HP, 0); noise = ao.make_noise(HP, B, 5)
lo = 9.21
res = {}
for mode in ("reference", "exact"):
    am.reset_default_graph()
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True, backward=mode, **HP)
    m.load_state_dict(params); m.set_noise(noise); m.set_dynamic(z_pres_prior_log_odds=lo)
    s = m._stream(); m._run_forward(s); m._run_backward(s); torch.cuda.synchronize()
    res[mode] = {k: v.detach().cpu().double().clone() for k, v in m.gradients.items()}
    res[mode + "_d"] = dict(d_sxyw=m.d_sxyw.cpu().double().clone(), d_out7=m.d_out7.cpu().double().clone())
for dt, name in ((torch.float32, "t32"), (torch.float64, "t64")):
    pt = at.to_torch(params, dtype=dt, requires_grad=True)
    out, grads = at.loss_and_grads(pt, torch.tensor(images, dtype=dt), torch.tensor(targets), at.to_torch(noise, dtype=dt), HP, lo)
    res[name] = {k: v.double() for k, v in grads.items()}
tot = {k: 0.0 for k in ("reference", "exact", "t32", "t64")}
for k in res["t64"]:
    row = [float(res[n][k].norm()) for n in ("reference", "exact", "t32", "t64")]
    for n, r in zip(("reference", "exact", "t32", "t64"), row): tot[n] += r * r
    print("%-36s hip_ref %.3e hip_exact %.3e t32 %.3e t64 %.3e" % (k, *row))
print({k: v ** 0.5 for k, v in tot.items()})
print("d_sxyw (write path) norms per step: ref", res["reference_d"]["d_sxyw"].norm(dim=(1,)).mean(0) if False else res["reference_d"]["d_sxyw"].abs().mean(dim=(0,1)), "exact", res["exact_d"]["d_sxyw"].abs().mean(dim=(0,1)))
print("d_out7 mean abs: ref", res["reference_d"]["d_out7"].abs().mean(dim=(0,1)), "\n exact", res["exact_d"]["d_out7"].abs().mean(dim=(0,1)))
