import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from air import _hip as H
from oracle import air_oracle_torch as at
rng = np.random.RandomState(0)
B, Cc, w = 8, 50, 28
vrec = rng.uniform(0.2, 0.8, (B, w * w)).astype(np.float32)
s = rng.uniform(0.25, 0.35, B).astype(np.float32); x = rng.uniform(-0.5, 0.5, B).astype(np.float32); y = rng.uniform(-0.5, 0.5, B).astype(np.float32)
img = np.zeros((B, 50, 50), np.float32)
for b in range(B):
    yy, xx = rng.randint(0, 30, 2); img[b, yy:yy + 18, xx:xx + 18] = rng.uniform(0.5, 1, (18, 18))
img = img.reshape(B, -1)
zval = 0.9
def torch_ref(dt):
    v = torch.tensor(vrec, dtype=dt).reshape(B, w, w)
    ss = torch.tensor(s, dtype=dt, requires_grad=True); xx_ = torch.tensor(x, dtype=dt, requires_grad=True); yy_ = torch.tensor(y, dtype=dt, requires_grad=True)
    z0 = torch.zeros_like(ss)
    th = torch.stack([torch.stack([1.0 / ss, z0, -xx_ / ss], 1), torch.stack([z0, 1.0 / ss, -yy_ / ss], 1)], 1)
    R = zval * at.transformer(v, th, (Cc, Cc)).reshape(B, -1)
    R.retain_grad()
    im = torch.tensor(img, dtype=dt)
    rc = torch.clamp(R, 0.0, 1.0)
    loss = -(im * torch.log(rc + 1e-9) + (1 - im) * torch.log(1 - rc + 1e-9)).sum()
    loss.backward()
    return R.detach(), R.grad.detach(), ss.grad, xx_.grad, yy_.grad
R32, g32, ds32, dx32, dy32 = torch_ref(torch.float32)
_, _, ds64, dx64, dy64 = torch_ref(torch.float64)
dev = "cuda"
att = torch.zeros(B, H.ATT_STRIDE, device=dev)
att[:, H.ATT_S] = torch.tensor(s); att[:, H.ATT_X] = torch.tensor(x); att[:, H.ATT_Y] = torch.tensor(y)
att[:, H.ATT_Z] = zval; att[:, H.ATT_MASK] = 1.0
vr = torch.tensor(vrec, device=dev); g = g32.to(dev).contiguous()
for lit in (0, 1):
    dgen = torch.zeros(B, w * w, device=dev); dsx = torch.zeros(B, 4, device=dev)
    wb = H.WriteBwd(C.c_void_p(g.data_ptr()), C.c_void_p(vr.data_ptr()), C.c_void_p(att.data_ptr()), C.c_void_p(dgen.data_ptr()),
                    C.c_void_p(dsx.data_ptr()), B, 1, Cc, w, lit)
    H.check(H.lib().air_write_bwd(C.byref(wb), None)); torch.cuda.synchronize()
    print("literal", lit)
    print("  ds hip  ", dsx[:, 0].cpu().numpy())
    print("  dx hip  ", dsx[:, 1].cpu().numpy())
print("  ds t32  ", ds32.numpy()); print("  ds t64  ", ds64.numpy())
print("  dx t32  ", dx32.numpy()); print("  dx t64  ", dx64.numpy())
print("  dy t32  ", dy32.numpy()); print("  dy t64  ", dy64.numpy())
