"""The numeric summaries of the reference (/root/reference/air/air_model.py:160-209, 608-625; evaluated by
training.py:169-200): oracle/summaries.py is their numpy restatement, air_summaries (csrc/air_summaries.hip) the one
launch the training driver uses.  CPU: the restatement on hand-computed cases, the name list, argument errors of the C
entry point.  GPU: the launch against the restatement on a real forward pass -- all steps reached, a loop that stopped
early (zero-padded columns, :187), an empty digit-count group (NaN, tf.reduce_mean of nothing)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import air_oracle as ao
from oracle import summaries as osum
from oracle.synth import blob_canvases


def test_names_follow_the_reference_order():
    names = osum.summary_names(3, 2)
    assert names[:2] == ["loss", "accuracy"]
    assert names[2:6] == ["steps_0_dig", "steps_1_dig", "steps_2_dig", "steps_all_dig"]          # :163-182
    assert names[6] == "rec_loss_0_dig" and names[14] == "total_loss_0_dig"                      # :614-617
    assert names[18] == "scale_1_step_0_dig" and names[22] == "scale_2_step_0_dig"               # :620, :196-208
    assert names[30] == "z_pres_prob_1_step_0_dig" and names[-1] == "vae_kl_3_step_all_dig"
    assert len(names) == 2 + (4 + 6 * 3) * 4 == len(set(names))
    from air import _hip as H
    assert H.lib().air_summaries_count(3, 2) == len(names)
    assert H.lib().air_summaries_count(5, 4) == len(osum.summary_names(5, 4))


def test_restatement_on_a_hand_computed_case():
    # 4 images, targets 0, 1, 1, 2; inferred counts 0, 1, 2, 2; the loop ran two of three steps (T' = 2)
    targets, digs = np.array([0, 1, 1, 2]), np.array([0, 1, 2, 2])
    stack = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0], [4.0, 40.0]], np.float32)
    v = osum.summaries(0.5, 0.75, targets, digs, [1, 2, 3, 4], [5, 6, 7, 8], stack, stack, stack, stack, stack, stack, 3, 2)
    d = dict(zip(osum.summary_names(3, 2), v))
    assert d["loss"] == 0.5 and d["accuracy"] == 0.75
    assert (d["steps_0_dig"], d["steps_1_dig"], d["steps_2_dig"], d["steps_all_dig"]) == (0.0, 1.5, 2.0, 1.25)
    assert (d["digit_acc_1_dig"], d["digit_acc_all_dig"]) == (0.5, 0.75)
    assert (d["rec_loss_1_dig"], d["total_loss_2_dig"]) == (2.5, 8.0)
    # step 1 of a masked quantity: images with steps > 0 (the last three); group 0 is empty -> NaN
    assert np.isnan(d["scale_1_step_0_dig"]) and d["scale_1_step_1_dig"] == 2.5 and d["scale_1_step_all_dig"] == 3.0
    # step 2: steps > 1 (the last two)
    assert d["scale_2_step_1_dig"] == 30.0 and d["scale_2_step_2_dig"] == 40.0 and d["scale_2_step_all_dig"] == 35.0
    # step 3 was never reached: zero padding, masked to steps > 2 -> nothing left
    assert np.isnan(d["scale_3_step_all_dig"])
    # one_more_step (z_pres_kl): step 1 keeps everything (steps > -1), step 3 keeps steps > 1 over the zero padding
    assert d["z_pres_kl_1_step_0_dig"] == 1.0 and d["z_pres_kl_1_step_all_dig"] == 2.5
    assert d["z_pres_kl_3_step_all_dig"] == 0.0 and d["z_pres_kl_3_step_2_dig"] == 0.0
    # all_steps (z_pres_prob): no mask at all; step 3 is the padding
    assert d["z_pres_prob_2_step_0_dig"] == 10.0 and d["z_pres_prob_3_step_all_dig"] == 0.0


def test_argument_errors_without_gpu():
    from air import _hip as H
    lib = H.lib()
    assert lib.air_summaries(None, None) == -1
    buf = (C.c_int64 * 8)()
    A = C.addressof(buf)
    mk = lambda **kw: H.Summaries(**dict(dict(att=A, targets=A, digits=A, rec_loss=A, loss_item=A, scalars=A, out=A,  # noqa: E731
                                              B=4, N=3, max_digits=2), **kw))
    assert lib.air_summaries(C.byref(mk(out=None)), None) == -1
    assert lib.air_summaries(C.byref(mk(B=0)), None) == -1
    assert lib.air_summaries(C.byref(mk(N=17)), None) == -2
    assert lib.air_summaries(C.byref(mk(max_digits=7)), None) == -2
    assert lib.air_batch_gather(None, None, None, None, None, 4, 8, None) == -1
    assert lib.air_batch_gather(A, A, A, A, A, 4, 6, None) == -3                     # rows are copied in 16-byte pieces
    assert lib.air_shuffle_batch_dequeue_many(None, 1, A, None) == -1


def _oracle_of(model, hp):
    np_ = lambda t: t.detach().cpu().numpy()   # noqa: E731
    return osum.summaries(float(model.loss), float(model.accuracy), np_(model.target_num_digits), np_(model.rec_num_digits),
                          np_(model.reconstruction_loss), np_(model.loss_per_item), np_(model.rec_scales[:, :, 0]),
                          np_(model.z_pres_probs), np_(model.z_pres_kls), np_(model.scale_kls), np_(model.shift_kls),
                          np_(model.vae_kls), hp["max_steps"], hp["max_digits"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,prec", [(1000, "fp32"), (64, "bf16"), (203, "fp32")])
def test_launch_equals_the_restatement(B, prec):
    from air import air_model as am
    hp = dict(ao.TRAINING_HP)
    images, targets = blob_canvases(B, hp["canvas_size"], hp["max_digits"], seed=11)
    am.reset_default_graph()
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=False,
                    scope="air", gemm_precision=prec, seed=3, **hp)
    m.set_dynamic(z_pres_prior_log_odds=0.5)
    m.forward()
    assert m.summary_names() == osum.summary_names(hp["max_steps"], hp["max_digits"])

    def check():
        got = m.numeric_summaries().cpu().numpy()
        ref = _oracle_of(m, hp)
        assert got.shape == ref.shape
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(got, ref, rtol=2e-7, atol=0, equal_nan=True)
        return dict(zip(m.summary_names(), got))
    d = check()
    assert m.steps_executed == 3 and np.isfinite(d["vae_kl_3_step_all_dig"])
    # a loop that stopped after its first step: T' = 1, the later columns are the zero padding of :187
    m.att[0, :, 11] = 0.0                                              # AIR_ATT_MASK of step 1: nobody is active after it
    m._steps_executed = None
    assert m.steps_executed == 1
    d = check()
    assert d["z_pres_prob_2_step_all_dig"] == 0.0 and d["z_pres_prob_1_step_all_dig"] > 0.0
    # an empty digit-count group
    m.target_num_digits.clamp_(max=1)
    d = check()
    assert np.isnan(d["steps_2_dig"]) and np.isnan(d["z_pres_prob_1_step_2_dig"]) and np.isfinite(d["steps_all_dig"])
    # caller-provided output vector; wrong size is an error, not a launch
    out = torch.zeros(len(d), device="cuda")
    assert m.numeric_summaries(out) is out and float(out[0]) == float(m.loss)
    with pytest.raises(ValueError):
        m.numeric_summaries(torch.zeros(5, device="cuda"))
