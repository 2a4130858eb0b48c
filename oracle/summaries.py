"""Test infrastructure only: the reference's numeric summaries restated in numpy.

/root/reference/air/air_model.py:160-182 (_summarize_by_digit_count: tf.reduce_mean of tf.boolean_mask per digit count
0 .. max_digits, then the mean over all images), :184-209 (_summarize_by_step: the [B, T'] stack zero-padded to
max_steps columns, column i masked by steps > i, steps > i - 1 with one_more_step, unmasked with all_steps) and the list
:608-625 in the order the reference appends it.  The product computes the same numbers in one launch (air_summaries,
include/air_hip.h); only tests compare the two.
"""
import numpy as np


def summary_names(max_steps, max_digits):
    names = ["loss", "accuracy"]

    def by_digit(name):
        names.extend("%s_%d_dig" % (name, i) for i in range(max_digits + 1))
        names.append(name + "_all_dig")
    for n in ("steps", "rec_loss", "digit_acc", "total_loss"):          # :614-617
        by_digit(n)
    for n in ("scale", "z_pres_prob", "z_pres_kl", "scale_kl", "shift_kl", "vae_kl"):   # :620-625
        for i in range(max_steps):
            by_digit("%s_%d_step" % (n, i + 1))
    return names


def _mean(v):
    v = np.asarray(v, np.float32)
    return np.float32(np.nan) if v.size == 0 else np.float32(v.astype(np.float64).mean())


def _by_digit_count(out, values, digits, max_digits):
    """:160-182"""
    values = np.asarray(values, np.float32)
    for i in range(max_digits + 1):
        out.append(_mean(values[digits == i]))
    out.append(_mean(values))


def _by_step(out, tensor, steps, targets, max_steps, max_digits, one_more_step=False, all_steps=False):
    """:184-209; tensor is [B, T']"""
    tensor = np.asarray(tensor, np.float32)
    tensor = np.pad(tensor, [(0, 0), (0, max_steps - tensor.shape[1])])
    for i in range(max_steps):
        if all_steps:
            _by_digit_count(out, tensor[:, i], targets, max_digits)
        else:
            mask = steps > (i - (1 if one_more_step else 0))
            _by_digit_count(out, tensor[:, i][mask], targets[mask], max_digits)


def summaries(loss, accuracy, targets, rec_num_digits, reconstruction_loss, loss_per_item,
              rec_scales0, z_pres_probs, z_pres_kls, scale_kls, shift_kls, vae_kls, max_steps, max_digits):
    """-> float32 vector in summary_names() order.  The [B, T'] stacks are the reference's outputs (T' <= max_steps)."""
    targets = np.asarray(targets)
    digs = np.asarray(rec_num_digits)
    out = [np.float32(loss), np.float32(accuracy)]
    _by_digit_count(out, digs.astype(np.float32), targets, max_digits)
    _by_digit_count(out, reconstruction_loss, targets, max_digits)
    _by_digit_count(out, (digs == targets).astype(np.float32), targets, max_digits)
    _by_digit_count(out, loss_per_item, targets, max_digits)
    _by_step(out, rec_scales0, digs, targets, max_steps, max_digits)
    _by_step(out, z_pres_probs, digs, targets, max_steps, max_digits, all_steps=True)
    _by_step(out, z_pres_kls, digs, targets, max_steps, max_digits, one_more_step=True)
    _by_step(out, scale_kls, digs, targets, max_steps, max_digits)
    _by_step(out, shift_kls, digs, targets, max_steps, max_digits)
    _by_step(out, vae_kls, digs, targets, max_steps, max_digits)
    return np.asarray(out, np.float32)
