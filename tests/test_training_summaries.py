"""Scalar summaries of the training driver against the reference's definitions
(air_model.py:160-209 `_summarize_by_digit_count`, `_summarize_by_step`, :614-625)."""
import math

import numpy as np
import torch


def _ref_by_digit_count(values, digits, name, max_digits):
    out = {}
    v = np.asarray(values, np.float64)
    for i in range(max_digits + 1):
        sel = v[digits == i]
        out["%s_%d_dig" % (name, i)] = float(sel.mean()) if len(sel) else float("nan")
    out[name + "_all_dig"] = float(v.mean()) if len(v) else float("nan")
    return out


def _ref_by_step(tensor, steps, digits, name, max_steps, max_digits, one_more_step=False, all_steps=False):
    t = np.zeros((tensor.shape[0], max_steps))                     # tf.pad to max_steps columns
    t[:, :tensor.shape[1]] = tensor
    out = {}
    for i in range(max_steps):
        if all_steps:
            out.update(_ref_by_digit_count(t[:, i], digits, "%s_%d_step" % (name, i + 1), max_digits))
        else:
            mask = steps > (i - (1 if one_more_step else 0))
            out.update(_ref_by_digit_count(t[:, i][mask], digits[mask], "%s_%d_step" % (name, i + 1), max_digits))
    return out


def test_summaries_match_reference_definitions():
    from training import Summaries
    rng = np.random.RandomState(0)
    B, T, N, D = 40, 2, 3, 2
    digits = rng.randint(0, D + 1, size=B)
    steps = rng.randint(0, N + 1, size=B)
    vals = rng.rand(B)
    per_step = rng.rand(B, T)                                      # fewer columns than max_steps
    sm = Summaries(torch.tensor(digits), D, N)
    sm.by_digit_count("rec_loss", torch.tensor(vals))
    sm.by_digit_count("digit_acc", torch.tensor(steps) == torch.tensor(digits))
    sm.by_step(torch.tensor(per_step), torch.tensor(steps), "scale")
    sm.by_step(torch.tensor(per_step), torch.tensor(steps), "z_pres_kl", one_more_step=True)
    sm.by_step(torch.tensor(per_step), torch.tensor(steps), "z_pres_prob", all_steps=True)
    got = sm.fetch()
    want = {}
    want.update(_ref_by_digit_count(vals, digits, "rec_loss", D))
    want.update(_ref_by_digit_count((steps == digits).astype(np.float64), digits, "digit_acc", D))
    want.update(_ref_by_step(per_step, steps, digits, "scale", N, D))
    want.update(_ref_by_step(per_step, steps, digits, "z_pres_kl", N, D, one_more_step=True))
    want.update(_ref_by_step(per_step, steps, digits, "z_pres_prob", N, D, all_steps=True))
    assert set(got) == set(want) and len(got) == (2 + 9) * (D + 2)
    for k, w in want.items():
        g = got[k]
        assert (math.isnan(g) and math.isnan(w)) or abs(g - w) < 1e-5, (k, g, w)
