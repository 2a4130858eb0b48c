"""Test infrastructure only (tests/, __graft_entry__.smoke()): where a whole-model ELBO difference comes from.

The loss of air_model.py:580-611 is  mean_b( BCE_b + sum_t KL_b,t ),  BCE_b = -sum_p x log(r + 1e-9) + (1 - x) log(1 - r + 1e-9)
(:580-593).  A canvas pixel outside every glimpse holds a rounding residue of the write transformer's out-of-range taps
-- 0 <= r < 1e-5, not zero -- and UNDER INK (x > 0) its term x log(r + 1e-9) moves by O(1) per pixel when r moves by 1e-7:
two fp32 evaluations that agree to 6.6e-6 in r differ by ~0.6 % in the ELBO through those pixels alone.  elbo_split()
separates that part from everything else, so that everything else can be held to 1e-4:

  bce_self     device reconstruction_loss against the fp64 BCE of the device's OWN reconstruction (the loss kernel)
  kl           mean_b sum_t KL, device against reference
  bce_regular  BCE difference summed over the pixels that are NOT residue pixels
  bce_residue  BCE difference summed over the residue pixels: x > 0 and min(r_device, r_reference) < 1e-5
  total        whole-model ELBO difference (= kl + bce_regular + bce_residue up to the fp32 rounding of the device's sums)
all relative to |reference ELBO|, signed except bce_self."""
import numpy as np

EPS = 1e-9
RESIDUE_BELOW = 1e-5


def _bce_pixels(x, r):
    return -(x * np.log(r + EPS) + (1.0 - x) * np.log(1.0 - r + EPS))


def elbo_split(images, dev_recon, dev_rec_loss, dev_loss_per_item, ref_recon, ref_rec_loss, ref_loss_per_item):
    x = np.asarray(images, np.float64)
    rd, ro = np.asarray(dev_recon, np.float64), np.asarray(ref_recon, np.float64)
    bd, bo = _bce_pixels(x, rd), _bce_pixels(x, ro)
    residue = (x > 0) & (np.minimum(rd, ro) < RESIDUE_BELOW)
    dl, rl = np.asarray(dev_rec_loss, np.float64), np.asarray(ref_rec_loss, np.float64)
    di, ri = np.asarray(dev_loss_per_item, np.float64), np.asarray(ref_loss_per_item, np.float64)
    scale = abs(ri.mean())
    diff = bd - bo
    return {
        "bce_self": float(np.abs(dl - bd.sum(1)).max() / max(1.0, np.abs(bd.sum(1)).max())),
        "kl": float(((di - dl) - (ri - rl)).mean() / scale),
        "bce_regular": float((diff * ~residue).sum(1).mean() / scale),
        "bce_residue": float((diff * residue).sum(1).mean() / scale),
        "total": float((di.mean() - ri.mean()) / scale),
        "residue_pixels": int(residue.sum()),
        "ink_pixels": int((x > 0).sum()),
    }


def check(split, regular_tol=1e-4, residue_tol=1e-2, self_tol=1e-5):
    """the statement the tests and smoke() make: everything but the residue pixels at 1e-4, those under the 1e-2 band"""
    assert split["bce_self"] <= self_tol, split
    assert abs(split["kl"]) + abs(split["bce_regular"]) <= regular_tol, split
    assert abs(split["bce_residue"]) <= residue_tol, split
    assert abs(split["total"]) <= residue_tol, split
    return split
