import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from air import air_model as am, _hip as H
import multi_mnist as mm
from bench import HP, ANNEAL

ds = mm.generate_dataset(2, 3000, 100)
dev = "cuda"
tr = torch.tensor(ds["train_images"], device=dev); td = torch.tensor(ds["train_digits"], device=dev)
B = 64
xin = torch.zeros(B, 2500, device=dev); tin = torch.zeros(B, dtype=torch.int32, device=dev)
am.reset_default_graph()
hp = dict(HP)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    hp[k] = float(v)
print("overrides", sys.argv[2:])
m = am.AIRModel(xin, tin, cnn=False, train=True, annealing_schedules=ANNEAL, **hp)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator(device=dev); g.manual_seed(0)
def show(it):
    a = m.att.cpu().numpy()
    vr = m.vrec.cpu().numpy(); rec = m.reconstruction.cpu().numpy(); x = xin.cpu().numpy()
    ink = (x * (rec > 1e-3)).sum() / max(x.sum(), 1e-9)
    print("it %d loss %.0f gn %.1f | s %.3f+-%.3f x %.2f+-%.2f y %.2f+-%.2f z %.2f mask %.2f | vrec mean %.3f max %.3f | recon max %.3f mean %.4f | ink covered %.3f | recloss %.0f" % (
        it, float(m.loss), float(m.store.gnorm), a[..., 0].mean(), a[..., 0].std(), a[..., 1].mean(), a[..., 1].std(), a[..., 2].mean(), a[..., 2].std(),
        a[..., 4].mean(), a[..., 11].mean(), vr.mean(), vr.max(), rec.max(), rec.mean(), ink, float(m.reconstruction_loss.mean())))
for it in range(steps + 1):
    idx = torch.randint(0, tr.shape[0], (B,), device=dev, generator=g)
    torch.index_select(tr, 0, idx, out=xin); torch.index_select(td, 0, idx, out=tin)
    m.training()
    if it % max(1, steps // 10) == 0: show(it)
# render one 1-digit image
x = xin.cpu().numpy(); rec = m.reconstruction.cpu().numpy(); t = tin.cpu().numpy(); a = m.att.cpu().numpy()
b = int(np.argmax(t == 1))
print("image", b, "targets", t[b], "s,x,y,z per step", a[:, b, :5])
for r in range(0, 50, 2):
    print(''.join('#' if v > .5 else ('+' if v > .1 else '.') for v in x[b].reshape(50, 50)[r]) + "   " + ''.join('#' if v > .5 else ('+' if v > .05 else '.') for v in rec[b].reshape(50, 50)[r]))
