import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from air import air_model as am
import multi_mnist as mm
from bench import HP, ANNEAL
mode = sys.argv[1]
ds = mm.generate_dataset(2, 3000, 100)
dev = "cuda"
tr = torch.tensor(ds["train_images"], device=dev); td = torch.tensor(ds["train_digits"], device=dev)
B = 64
xin = torch.zeros(B, 2500, device=dev); tin = torch.zeros(B, dtype=torch.int32, device=dev)
am.reset_default_graph()
hp = dict(HP); hp["learning_rate"] = 1e-3
m = am.AIRModel(xin, tin, cnn=False, train=True, annealing_schedules=ANNEAL, **hp)
g = torch.Generator(device=dev); g.manual_seed(0)
for it in range(1201):
    idx = torch.randint(0, tr.shape[0], (B,), device=dev, generator=g)
    torch.index_select(tr, 0, idx, out=xin); torch.index_select(td, 0, idx, out=tin)
    if mode == "torchnoise":
        m.eps_scale.normal_(generator=g); m.eps_shift.normal_(generator=g); m.eps_z.normal_(generator=g); m.eps_x.normal_(generator=g); m.u.uniform_(generator=g)
        m._injected_noise = True
    m.training()
    if it % 200 == 0:
        a = m.att.cpu().numpy(); rec = m.reconstruction.cpu().numpy(); x = xin.cpu().numpy()
        ink = (x * (rec > 1e-3)).sum() / max(x.sum(), 1e-9)
        n = m.normals
        print("%s it %d loss %.0f gn %.1f | s %.3f+-%.3f x %.2f+-%.2f | ink covered %.3f | recloss %.0f | noise mean %.4f std %.4f u mean %.3f" % (
            mode, it, float(m.loss), float(m.store.gnorm), a[..., 0].mean(), a[..., 0].std(), a[..., 1].mean(), a[..., 1].std(), ink,
            float(m.reconstruction_loss.mean()), float(n.mean()), float(n.std()), float(m.uniforms.mean())))
