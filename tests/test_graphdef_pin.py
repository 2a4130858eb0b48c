"""Pins the oracle's TF-internal assumptions to the reference's own serialized training graph.

tests/golden/graphdef_facts.json is DATA extracted from /root/reference/model/air-model.meta
(the MetaGraphDef TF 1.3 wrote after 270k iterations) by oracle/graphdef_pin.py: node wiring,
scalar constants, initializer limits.  These are the only artefacts of the reference's
*executed* arithmetic available offline (no TensorFlow, no reference tests); every fact the
Python sources do not show is asserted here against the oracle restatement.
"""
import inspect
import json
import math
import os

import numpy as np
import pytest

from oracle import air_oracle as O

FACTS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "graphdef_facts.json")))


def test_graph_is_the_tf13_training_graph():
    assert FACTS["tensorflow_version"] == "1.3.0"
    assert FACTS["apply_adam_count"] == len(O.param_shapes(O.TRAINING_HP)) == 36
    assert FACTS["lstm_split_num"] == 4


def _run_cell_graph(gates, c):
    """Interprets the recorded wiring of air/rnn/while/rnn/* (Split, Add, Mul, Sigmoid, Tanh)."""
    w = FACTS["lstm_cell_wiring"]
    consts = {"rnn/add/y": np.float32(FACTS["body_scalar_consts"]["rnn/add/y"])}
    parts = np.split(gates, FACTS["lstm_split_num"], axis=1)
    memo = {}

    def ev(name):
        if name in memo:
            return memo[name]
        if name == "Identity_2":            # loop variable: cell state (the concat takes Identity_3 = h)
            return c
        if name in consts:
            return consts[name]
        if name.startswith("rnn/split"):
            return parts[int(name.split(":")[1]) if ":" in name else 0]
        op, *ins = w[name]
        a = [ev(i) for i in ins]
        r = {"Add": lambda: a[0] + a[1], "Mul": lambda: a[0] * a[1],
             "Sigmoid": lambda: O.sigmoid(a[0]), "Tanh": lambda: np.tanh(a[0])}[op]()
        memo[name] = r
        return r

    return ev("rnn/add_1"), ev("rnn/mul_2")          # c', h'


def test_lstm_cell_matches_recorded_wiring():
    """Gate order i, j, f, o and forget bias 1.0: the oracle cell must equal
    the graph's own pointwise wiring evaluated on the same pre-activations."""
    assert FACTS["lstm_ops"]["rnn/rnn_1/concat"][0] == "ConcatV2"
    assert FACTS["lstm_ops"]["rnn/rnn_1/concat"][2].endswith("Identity_3")      # [x, h]
    rng = np.random.RandomState(0)
    B, D, R = 5, 7, 6
    x = rng.randn(B, D).astype(np.float32)
    h = rng.randn(B, R).astype(np.float32)
    c = rng.randn(B, R).astype(np.float32)
    K = rng.randn(D + R, 4 * R).astype(np.float32)
    b = rng.randn(4 * R).astype(np.float32)
    gates = np.concatenate([x, h], axis=1) @ K + b
    c_ref, h_ref = _run_cell_graph(gates, c)
    c_new, h_new = O.lstm_cell(x, c, h, K, b)
    np.testing.assert_array_equal(c_new, c_ref)
    np.testing.assert_array_equal(h_new, h_ref)


def test_glorot_limits_match_recorded_initializers():
    shapes = O.param_shapes(O.TRAINING_HP)
    rec = FACTS["initializer_uniform_max"]
    two_d = {k: v for k, v in shapes.items() if len(v) == 2}
    assert len(rec) == len(two_d) == 18
    for name, (fi, fo) in two_d.items():
        key = "air/rnn/" + ("rnn/kernel" if name == "rnn/kernel" else name)
        assert rec[key] == pytest.approx(math.sqrt(6.0 / (fi + fo)), rel=1e-6), name
    p = O.init_params(O.TRAINING_HP, seed=3)
    for name, (fi, fo) in two_d.items():
        lim = math.sqrt(6.0 / (fi + fo))
        assert np.abs(p[name]).max() <= lim and np.abs(p[name]).max() > 0.9 * lim
    for name, s in shapes.items():
        if len(s) == 1:
            assert not p[name].any()


def test_activations_match_recorded_ops():
    acts = FACTS["body_activations"]
    for i in (1, 2):
        assert acts["vae/recognition_%d/recognition_%d/Softplus" % (i, i)] == "Softplus"
        assert acts["vae/generative_%d/generative_%d/Softplus" % (i, i)] == "Softplus"
    for head in ("scale/mean", "scale/log_variance", "shift/mean", "shift/log_variance", "z_pres/log_odds"):
        assert acts[head + "/hidden/hidden/Relu"] == "Relu"
    assert acts["scale/Sigmoid"] == "Sigmoid" and acts["shift/Tanh"] == "Tanh"
    assert acts["vae/gen_sample/Sigmoid"] == "Sigmoid"
    # no activation on the output layers: the only Relu/Softplus nodes are the hidden ones above
    assert sum(1 for v in acts.values() if v == "Relu") == 5
    assert sum(1 for v in acts.values() if v == "Softplus") == 4
    # the unseeded RNG ops of the body: 4 normals + the Concrete uniform (make_noise protocol)
    ops = sorted(v[0] for v in FACTS["rng_ops"].values())
    assert ops == ["RandomStandardNormal"] * 4 + ["RandomUniform"]
    assert all(v[1] == 0 and v[2] == 0 for v in FACTS["rng_ops"].values())     # seed = seed2 = 0: unseeded


def test_scalar_constants_match_oracle():
    sc = FACTS["body_scalar_consts"]
    hp = O.TRAINING_HP
    f32 = lambda v: float(np.float32(v))
    assert sc["Less_1/y"] == f32(hp["stopping_threshold"])
    assert sc["loss/z_pres_kl/add_3/y"] == f32(O.EPS) == f32(1e-9)
    assert sc["loss/scale_kl/truediv/y"] == f32(hp["scale_prior_variance"])
    assert sc["loss/scale_kl/sub_2/y"] == f32(hp["scale_prior_mean"])
    assert sc["loss/shift_kl/truediv/y"] == f32(hp["shift_prior_variance"])
    assert sc["loss/VAE_kl/truediv/y"] == f32(hp["vae_prior_variance"])
    assert sc["rnn/add/y"] == 1.0
    for d in ("st_forward", "st_backward"):
        base = d + "/SpatialTransformer/_transform/_interpolate/"
        assert sc[base + "sub_2/y"] == sc[base + "sub_3/y"] == f32(1.001)
        assert sc[base + "truediv/y"] == 2.0
        assert sc[d + "/SpatialTransformer/_transform/_meshgrid/LinSpace/start"] == -1.0
        assert sc[d + "/SpatialTransformer/_transform/_meshgrid/LinSpace/stop"] == 1.0
    assert "1.001" in inspect.getsource(O.transformer)


def test_optimizer_and_annealing_constants_match_oracle():
    tr = FACTS["training_scalar_consts"]
    hp = O.TRAINING_HP
    f32 = lambda v: float(np.float32(v))
    sig = inspect.signature(O.adam_step).parameters
    assert tr["air/training/Adam/beta1"] == f32(sig["beta1"].default)
    assert tr["air/training/Adam/beta2"] == f32(sig["beta2"].default)
    assert tr["air/training/Adam/epsilon"] == f32(sig["epsilon"].default)
    assert tr["air/training/Adam/learning_rate"] == f32(hp["learning_rate"])
    assert tr["air/training/clip_by_global_norm/Const"] == f32(hp["gradient_clipping_norm"])
    an = O.TRAINING_ANNEALING["z_pres_prior_log_odds"]
    assert tr["air/z_pres_prior_log_odds/learning_rate"] == f32(an["init"])
    assert tr["air/z_pres_prior_log_odds/Cast_2/x"] == f32(an["factor"])
    assert tr["air/z_pres_prior_log_odds_max/y"] == f32(an["min"])
    # one UnsortedSegmentSum in the whole graph: the dense conversion of the sampler's Gather
    # gradients happens once per gathered tensor, after the IndexedSlices of the taps were
    # concatenated -- the scatter order the HIP "reference" backward reproduces
    assert FACTS["unsorted_segment_sum_count"] >= 1
