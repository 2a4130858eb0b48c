"""Gradient parity (HIP vs fp64 oracle) at a TRAINED state, exercising masks/saturation branches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from air import air_model as am
from oracle import air_oracle as ao, air_oracle_torch as at
import multi_mnist as mm

HP = dict(ao.TRAINING_HP)
ds = mm.generate_dataset(2, 3000, 100)
dev = "cuda"
tr = torch.tensor(ds["train_images"], device=dev); td = torch.tensor(ds["train_digits"], device=dev)
B = 64
xin = torch.zeros(B, 2500, device=dev); tin = torch.zeros(B, dtype=torch.int32, device=dev)
am.reset_default_graph()
m = am.AIRModel(xin, tin, cnn=False, train=True, annealing_schedules=ao.TRAINING_ANNEALING, **HP)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g = torch.Generator(device=dev); g.manual_seed(0)
for it in range(steps):
    idx = torch.randint(0, tr.shape[0], (B,), device=dev, generator=g)
    torch.index_select(tr, 0, idx, out=xin); torch.index_select(td, 0, idx, out=tin)
    m.training()
    if it % 2000 == 0: print(it, float(m.loss), float(m.accuracy), float(m.store.gnorm))
torch.cuda.synchronize()
params = {k: v.numpy() for k, v in m.state_dict().items() if not k.startswith("_") and k != "global_step"}
noise = ao.make_noise(HP, B, 5)
m.set_noise(noise)
lo = float(m.dyn[0])
s = m._stream(); m._run_forward(s); m._run_backward(s); torch.cuda.synchronize()
images, targets = xin.cpu().numpy(), tin.cpu().numpy()
o32 = ao.air_forward(params, images, targets, noise, HP, True, lo)
print("prior_lo", lo, "loss hip", float(m.loss), "oracle32", float(o32["loss"]), "digits eq", (m.rec_num_digits.cpu().numpy() == o32["rec_num_digits"]).mean())
print("recon maxdiff", np.abs(m.reconstruction.cpu().numpy() - o32["reconstruction"]).max())
print("z", o32["_z_pres"][:4])
f64 = torch.float64
pt = at.to_torch(params, dtype=f64, requires_grad=True)
out, grads = at.loss_and_grads(pt, torch.tensor(images, dtype=f64), torch.tensor(targets), at.to_torch(noise, dtype=f64), HP, lo)
print("loss64", float(out["loss"]))
for k, gref in grads.items():
    got = m.gradients[k].detach().cpu().double(); ref = gref.double()
    print("%-40s |ref| %.3e |got| %.3e rel %.3e" % (k, float(ref.norm()), float(got.norm()), float((got - ref).norm() / max(float(ref.norm()), 1e-30))))
