"""GPU parity tests of the whole hot path (AIRModel on libair_hip.so) against the
CPU oracle: forward outputs, ELBO, all 36 gradients, the clipped Adam update.

Stated tolerances (fp32 kernels vs fp32 oracle, identical injected noise):
  reconstruction |d| <= 2e-5; per-step KLs rel 1e-4; rec_num_digits exact;
  BCE given the SAME reconstruction rel 1e-5; ELBO: KL terms + every non-residue pixel rel 1e-4, the
  residue pixels (0 <= r < 1e-5 under ink: out-of-range sampler residues through log(r + 1e-9))
  alone under the rel 1e-2 band (oracle/elbo_split.py);
  gradients vs the fp64 evaluation of the same graph: per-tensor relative L2 error <= 5e-3
  (the fp32 autograd of the reference formulation is noise-dominated, see _grad_check).
bf16-GEMM path: compared with the same oracle at looser, measured tolerances."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import air_oracle as ao  # noqa: E402
from oracle import air_oracle_torch as at  # noqa: E402
from oracle import elbo_split  # noqa: E402
from oracle.synth import blob_canvases  # noqa: E402

HP = dict(ao.TRAINING_HP)
REPORT = {}


@pytest.fixture(scope="module")
def am():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from air import air_model
    return air_model


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _make(am, B, train, seed_img=3, seed_noise=1, prec="fp32", scope=None, lo=-2.0, hp=HP, blank=False,
          backward="exact", **ctor_kw):
    images, targets = blob_canvases(B, hp["canvas_size"], hp["max_digits"], seed=seed_img)
    noise = None
    if blank:
        # smooth regime: no ink (no log(r) pole at r -> 0) and z_pres ~ 0.55 so that the canvas
        # stays well below 1 (no log(1 - r) pole at r -> 1)
        images = np.zeros_like(images)
        noise = ao.make_noise(hp, B, seed_noise)
        noise["u"][:] = 0.55
    params = ao.init_params(hp, 0)
    if noise is None:
        noise = ao.make_noise(hp, B, seed_noise)
    am.reset_default_graph()
    model = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"),
                        cnn=False, train=train, scope=scope or "air", gemm_precision=prec, backward=backward, **ctor_kw, **hp)
    model.load_state_dict(params)
    model.set_noise(noise)
    model.set_dynamic(z_pres_prior_log_odds=lo)
    return model, images, targets, params, noise


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("B", [4, 64])
@pytest.mark.parametrize("train", [True, False])
def test_forward_parity_fp32(am, B, train):
    model, images, targets, params, noise = _make(am, B, train)
    model.forward()
    torch.cuda.synchronize()
    o = ao.air_forward(params, images, targets, noise, HP, train, -2.0, early_exit=True)
    T = o["steps_executed"]
    assert model.steps_executed == T
    rep = {}
    rep["recon"] = float(np.abs(_np(model.reconstruction) - o["reconstruction"]).max())
    assert rep["recon"] <= 2e-5
    assert np.array_equal(_np(model.rec_num_digits), o["rec_num_digits"])
    for k in ("rec_scales", "rec_shifts", "rec_windows", "rec_latents", "rec_st_back", "z_pres_probs"):
        got = _np(getattr(model, k))
        assert got.shape == o[k].shape, (k, got.shape, o[k].shape)
        rep[k] = float(np.abs(got - o[k]).max())
        assert rep[k] <= 5e-5 * max(1.0, np.abs(o[k]).max()), (k, rep[k])
    for k in ("z_pres_kls", "scale_kls", "shift_kls", "vae_kls"):
        got = _np(getattr(model, k))
        rep[k] = float(np.abs(got - o[k]).max() / max(1.0, np.abs(o[k]).max()))
        assert rep[k] <= 1e-4, (k, rep[k])
    # BCE on the device's own reconstruction, recomputed in fp64 on the host
    r = _np(model.reconstruction).astype(np.float64)
    x = images.astype(np.float64)
    bce = -np.sum(x * np.log(r + ao.EPS) + (1 - x) * np.log(1 - r + ao.EPS), axis=1)
    np.testing.assert_allclose(_np(model.reconstruction_loss), bce, rtol=1e-5, atol=1e-4)
    rep["elbo_rel"] = abs(float(model.loss) - float(o["loss"])) / abs(float(o["loss"]))
    # where the ELBO difference comes from (oracle/elbo_split.py): the KL terms and every pixel that is not an
    # out-of-range residue under ink at 1e-4 of the ELBO; only the residue pixels' log(r + 1e-9) under the 1e-2 band
    sp = elbo_split.elbo_split(images, _np(model.reconstruction), _np(model.reconstruction_loss), _np(model.loss_per_item),
                               o["reconstruction"], o["reconstruction_loss"], o["loss_per_item"])
    rep["elbo_split"] = sp
    print("ELBO split B=%d train=%s: %r" % (B, train, sp))
    elbo_split.check(sp)
    assert rep["elbo_rel"] <= 1e-2
    assert abs(float(model.accuracy) - float(o["accuracy"])) < 1e-6
    REPORT["forward_fp32_B%d_%s" % (B, "train" if train else "test")] = rep


def test_golden_fixture(am, golden_dir):
    g = np.load(os.path.join(golden_dir, "air_b4.npz"))
    lo0 = float(ao.annealed_value(ao.TRAINING_ANNEALING["z_pres_prior_log_odds"], 0))
    for tag, train, lo in (("train_lo9", True, lo0), ("train_lom2", True, -2.0), ("test_lom2", False, -2.0)):
        model, *_ = _make(am, 4, train, lo=lo)
        model.forward()
        assert np.array_equal(_np(model.rec_num_digits), g[tag + "/rec_num_digits"])
        assert np.abs(_np(model.reconstruction) - g[tag + "/reconstruction"]).max() <= 2e-5
        T = model.steps_executed
        for k in ("z_pres_kls", "scale_kls", "shift_kls", "vae_kls"):
            np.testing.assert_allclose(_np(getattr(model, k)), g[tag + "/" + k][:, :T], rtol=2e-4, atol=2e-4)
        assert abs(float(model.loss) - float(g[tag + "/loss"])) / abs(float(g[tag + "/loss"])) <= 1e-2


def _grad_check(am, B, prec, tol, emulate_bf16=False, blank=False, tag="", loose=("z_pres", "rnn")):
    model, images, targets, params, noise = _make(am, B, True, prec=prec, blank=blank)
    # run forward+backward only (no optimizer): use the programs directly
    s = model._stream()
    model._run_forward(s)
    model._run_backward(s)
    torch.cuda.synchronize()
    # backward="exact": the HIP backward with out-of-range taps pre-merged (exact adjoint) must
    # match the fp64 evaluation of the same graph.  (The fp32 autodiff of the reference
    # formulation differs from it by O(10x) in norm: out-of-range taps scatter +/-w*g pairs with
    # g ~ 1e9/B that cancel only to rounding -- |g|_fp32 = 8372 vs |g|_fp64 = 622 at B=16.  That
    # residue is reproduced by backward="reference", tested separately below.)
    f64 = torch.float64
    at.MATMUL_MODE = "bf16" if emulate_bf16 else "exact"
    try:
        pt = at.to_torch(params, dtype=f64, requires_grad=True)
        _, grads = at.loss_and_grads(pt, torch.tensor(images, dtype=f64), torch.tensor(targets),
                                     at.to_torch(noise, dtype=f64), HP, -2.0)
    finally:
        at.MATMUL_MODE = "exact"
    rep = {}
    for k, gref in grads.items():
        got = model.gradients[k].detach().cpu().double()
        ref = gref.double()
        rep[k] = float((got - ref).norm() / max(ref.norm(), 1e-12))
    REPORT["grads_%s_B%d%s%s" % (prec, B, "_emul" if emulate_bf16 else "", tag)] = rep
    if blank:
        rep.pop("rnn/kernel")       # x == 0: only the recurrent rows carry gradient; covered by rnn/bias
    if tol is not None:
        bad = {k: v for k, v in rep.items() if v > (tol * 10 if loose and k.startswith(loose) else tol)}
        assert not bad, bad
    return model, grads


@pytest.mark.parametrize("B,tol", [(4, 1e-2), (64, 2e-3)])
def test_gradients_fp32(am, B, tol):
    # B=64: 1e-3 everywhere, 1e-2 for z_pres/* and rnn/* whose d z_pres = sum(dR * window_recon)
    # inherits fp32 forward round-off amplified by 1/(r + 1e-9); B=4 has fewer items to average
    _grad_check(am, B, "fp32", tol)


def test_gradients_bf16(am):
    """bf16-operand GEMMs (fwd, dgrad, wgrad), the arithmetic bench.py runs.  The Bernoulli ELBO has
    poles at r -> 0 under ink and r -> 1 off ink (d log(r + 1e-9), d log(1 - r + 1e-9)); at
    initialisation a few such pixels dominate the gradient, so the ~1e-3 perturbation bf16 makes to the
    canvas changes it by O(1) relative to an evaluation with exact products (measured 0.5-1.1 on the
    where-heads: reported, not asserted).  Against the fp64 twin that rounds the SAME GEMM operands to
    bf16 (oracle/air_oracle_torch.MATMUL_MODE = "bf16": every MatMul and both of its gradient MatMuls,
    the seven head output units exact as attend_fwd/attend_bwd compute them) the inked gradients ARE
    assertable: per-tensor relative L2 <= 1e-2 on all 36 variables (measured <= 4.7e-3), and <= 5e-4 in
    the smooth regime (blank canvases, z_pres ~ 0.55; measured <= 1.4e-4)."""
    _grad_check(am, 64, "bf16", None, tag="_inked")
    _grad_check(am, 64, "bf16", 1e-2, emulate_bf16=True, tag="_inked", loose=())
    _grad_check(am, 64, "bf16", 5e-4, emulate_bf16=True, blank=True, tag="_blank", loose=())
    _grad_check(am, 64, "bf16", 5e-2, blank=True, tag="_blank")
    _grad_check(am, 64, "fp32", 1e-3, blank=True, tag="_blank")


def test_forward_parity_bf16(am):
    model, images, targets, params, noise = _make(am, 64, True, prec="bf16")
    model.forward()
    o = ao.air_forward(params, images, targets, noise, HP, True, -2.0)
    rep = dict(recon=float(np.abs(_np(model.reconstruction) - o["reconstruction"]).max()),
               elbo_rel=abs(float(model.loss) - float(o["loss"])) / abs(float(o["loss"])),
               digits_equal=float((_np(model.rec_num_digits) == o["rec_num_digits"]).mean()))
    REPORT["forward_bf16_B64"] = rep
    assert rep["recon"] <= 3e-2 and rep["elbo_rel"] <= 3e-2 and rep["digits_equal"] >= 0.95


def test_train_step_matches_oracle_update(am):
    B = 16
    model, images, targets, params, noise = _make(am, B, True)
    model.training()
    torch.cuda.synchronize()
    assert int(model.global_step) == 1
    f64 = torch.float64
    pt = at.to_torch(params, dtype=f64, requires_grad=True)
    out, grads = at.loss_and_grads(pt, torch.tensor(images, dtype=f64), torch.tensor(targets),
                                   at.to_torch(noise, dtype=f64), HP, -2.0)
    m = {k: torch.zeros_like(p) for k, p in pt.items()}
    v = {k: torch.zeros_like(p) for k, p in pt.items()}
    gn = at.clip_and_adam(pt, grads, m, v, 1, HP)
    assert abs(float(model.store.gnorm) - float(gn)) / float(gn) < 2e-3
    worst = 0.0
    for k, p in pt.items():
        got = model.variables[k].detach().cpu().double()
        # first Adam step moves every weight by ~lr: compare the DELTA
        d_ref = (p.detach() - torch.as_tensor(params[k])).double()
        d_got = (got - torch.as_tensor(params[k])).double()
        err = float((d_got - d_ref).norm() / max(d_ref.norm(), 1e-12))
        worst = max(worst, err)
        assert err < 1e-3, (k, err)      # measured worst 7.6e-5 (the first Adam step is sign-like: ~lr per entry)
    REPORT["adam_delta_worst_rel"] = worst
    o32 = ao.air_forward(params, images, targets, noise, HP, True, -2.0)
    assert abs(float(model.loss) - float(o32["loss"])) / abs(float(o32["loss"])) < 1e-2


def test_determinism_and_graph_replay(am):
    model, *_ = _make(am, 32, True)
    sd = model.state_dict()
    model.training()
    torch.cuda.synchronize()
    p1 = model.store.params.clone()
    l1 = float(model.loss)
    model.load_state_dict(sd)
    model.training()
    torch.cuda.synchronize()
    assert torch.equal(p1, model.store.params)       # atomics-free: bit-identical
    assert l1 == float(model.loss)


def test_variable_sharing_and_test_model(am):
    am.reset_default_graph()
    B = 8
    images, targets = blob_canvases(B, 50, 2, seed=9)
    dev = "cuda"
    tr = am.AIRModel(torch.tensor(images, device=dev), torch.tensor(targets, device=dev), cnn=False,
                     train=True, scope="air", **HP)
    te = am.AIRModel(torch.tensor(images, device=dev), torch.tensor(targets, device=dev), cnn=False,
                     train=False, reuse=True, scope="air", **HP)
    assert te.store is tr.store
    with pytest.raises(ValueError):
        am.AIRModel(torch.tensor(images, device=dev), torch.tensor(targets, device=dev), cnn=False, scope="air", **HP)
    with pytest.raises(NotImplementedError):
        am.AIRModel(torch.tensor(images, device=dev), torch.tensor(targets, device=dev), scope="other", **HP)
    for _ in range(3):
        tr.training()
    te.forward()
    torch.cuda.synchronize()
    assert int(te.global_step) == 3
    z = te.att[:, :, 4]
    assert bool(((z == 0) | (z == 1)).all())          # test mode rounds z_pres (reference :389-390)
    assert np.isfinite(float(te.loss))


def test_annealed_training_runs_and_loss_drops(am):
    am.reset_default_graph()
    B = 64
    images, targets = blob_canvases(B, 50, 2, seed=21)
    model = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False,
                        train=True, scope="air", annealing_schedules=ao.TRAINING_ANNEALING, **HP)
    losses = []
    for i in range(60):
        model.training()
        if i % 10 == 0 or i == 59:
            losses.append(float(model.loss))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    assert abs(float(model.dyn[0]) - float(ao.annealed_value(ao.TRAINING_ANNEALING["z_pres_prior_log_odds"], 59))) < 2e-3
    REPORT["train60_losses"] = losses


def test_reference_backward_carries_a_residue_the_exact_adjoint_does_not(am):
    """backward="reference" evaluates the sampler gradients in the op order of the reference's saved graph (pinned bit for
    bit in tests/test_gpu_graph_golden.py).  At out-of-range taps the +/- pairs cancel only to rounding and are multiplied
    by g ~ 1/(r + 1e-9): the where-heads, the LSTM and the VAE decoder then receive gradients far above the exact ones
    (backward="exact"), while the paths without out-of-range taps are unaffected."""
    B = 64
    import multi_mnist as mm
    ds = mm.generate_dataset(2, 100, 10)
    images, targets = ds["train_images"][:B], ds["train_digits"][:B]
    params, noise = ao.init_params(HP, 0), ao.make_noise(HP, B, 5)
    norms = {}
    for mode in ("reference", "exact"):
        am.reset_default_graph()
        m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False,
                        train=True, backward=mode, **HP)
        m.load_state_dict(params)
        m.set_noise(noise)
        m.set_dynamic(z_pres_prior_log_odds=9.21)
        s = m._stream()
        m._run_forward(s)
        m._run_backward(s)
        torch.cuda.synchronize()
        norms[mode] = {k: float(v.norm()) for k, v in m.gradients.items()}
    REPORT["residue_norms"] = {k: [norms["reference"][k], norms["exact"][k]] for k in norms["exact"]}
    inflated = [k for k in norms["exact"] if norms["reference"][k] > 2.0 * norms["exact"][k]]
    assert len(inflated) >= 3, inflated
    assert norms["reference"]["rnn/kernel"] > 1.15 * norms["exact"]["rnn/kernel"]
    for k in ("z_pres/log_odds/output/weights",):       # paths without out-of-range taps are unaffected
        assert abs(norms["reference"][k] - norms["exact"][k]) / norms["exact"][k] < 0.2


def test_reference_backward_learns_exact_does_not(am):
    """Behavioural parity with the reference: with its fp32 gradient semantics the model learns
    to attend and reconstruct within ~1000 steps (lr 1e-3); with the mathematically exact
    gradient it does not (neither does the fp64 oracle twin) -- DESIGN.md section 2."""
    import multi_mnist as mm
    ds = mm.generate_dataset(2, 1500, 10)
    dev = "cuda"
    tr = torch.tensor(ds["train_images"], device=dev)
    td = torch.tensor(ds["train_digits"], device=dev)
    hp = dict(HP, learning_rate=1e-3)
    out = {}
    for mode in ("reference", "exact"):
        am.reset_default_graph()
        xin = torch.zeros(64, 2500, device=dev)
        tin = torch.zeros(64, dtype=torch.int32, device=dev)
        m = am.AIRModel(xin, tin, cnn=False, train=True, annealing_schedules=ao.TRAINING_ANNEALING,
                        backward=mode, **hp)
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        rec = []
        for it in range(1500):
            idx = torch.randint(0, tr.shape[0], (64,), device=dev, generator=g)
            torch.index_select(tr, 0, idx, out=xin)
            torch.index_select(td, 0, idx, out=tin)
            m.training()
            if it >= 1400:
                rec.append(float(m.reconstruction_loss.mean()))
        out[mode] = float(np.mean(rec))
    REPORT["rec_loss_after_1500_steps"] = out
    assert out["reference"] < 600.0, out
    assert out["exact"] > 2.0 * out["reference"], out


def test_reference_and_exact_backward_agree_without_residues(am):
    """Structural check of backward="reference": on blank canvases with z ~ 0.55 (no log(r) pole, so
    d loss / d canvas stays O(1) and the fp32 residue of the out-of-range taps is ~1e-7 of it) the
    reference-order backward must equal the exact adjoint to rounding, for every variable."""
    grads = {}
    for mode in ("exact", "reference", "reference_carried"):
        model, *_ = _make(am, 64, True, blank=True, backward=mode)
        s = model._stream()
        model._run_forward(s)
        model._run_backward(s)
        torch.cuda.synchronize()
        grads[mode] = {k: v.detach().cpu().double().clone() for k, v in model.gradients.items()}
    worst = 0.0
    for k, ge in grads["exact"].items():
        for mode in ("reference", "reference_carried"):
            err = float((grads[mode][k] - ge).norm() / (ge.norm() + 1e-30))
            worst = max(worst, err)
            assert err <= 2e-3, (mode, k, err)
    REPORT["reference_vs_exact_smooth_regime_worst_rel"] = worst


def test_inference_graph_replay_equals_eager(am):
    """train=False models capture the whole forward as ONE hipGraph (the demo's hot call,
    demo/model_wrapper.py:19-30); a replay is bit-identical to the eager launches."""
    images, targets = blob_canvases(64, HP["canvas_size"], HP["max_digits"], seed=9)
    am.reset_default_graph()
    tr = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False,
                     train=True, scope="air", **HP)
    inf = am.AIRModel(tr.input_images, tr.target_num_digits, cnn=False, train=False, reuse=True, scope="air", **HP)
    inf.forward()
    torch.cuda.synchronize()
    eager = {k: _np(getattr(inf, k)).copy() for k in ("reconstruction", "rec_num_digits", "rec_scales", "rec_windows",
                                                     "reconstruction_loss", "vae_kls")}
    l0 = float(inf.loss)
    inf.capture_graph()
    inf._recon.zero_()
    inf.forward()
    torch.cuda.synchronize()
    for k, v in eager.items():
        assert np.array_equal(_np(getattr(inf, k)), v), k
    assert float(inf.loss) == l0
    # the captured graph follows the shared variables: a train step changes the next replay
    tr.training()
    inf.forward()
    torch.cuda.synchronize()
    assert float(inf.loss) != l0


def test_first_lstm_step_in_the_xwx_launch_matches_the_default_path(am, monkeypatch):
    """AIR_STEP0_FUSION=1 (x.Wx un-split with the zero-state first LSTM step in its epilogue, AIR_EPI_LSTM_FWD0)
    gives the same forward as the default split-K product + pointwise first step, up to the summation order
    of the K = 2500 contraction."""
    outs = {}
    for fused in (False, True):
        if fused:
            monkeypatch.setenv("AIR_STEP0_FUSION", "1")
        else:
            monkeypatch.delenv("AIR_STEP0_FUSION", raising=False)
        model, *_ = _make(am, 16, True, blank=True)
        assert model._fuse_step0 == fused
        model.forward()
        torch.cuda.synchronize()
        outs[fused] = (model.h.clone(), model.c.clone(), model.acts.clone(), float(model.loss))
    for a, b in zip(outs[False][:3], outs[True][:3]):
        assert (a - b).abs().max() < 2e-5
    assert abs(outs[False][3] - outs[True][3]) / abs(outs[False][3]) < 1e-5


def _assert_panels_current(st):
    """the panel-blocked twins on the store hold bf16 of the CURRENT variables in air_panel_t's layout"""
    pan = st.params16p.cpu().numpy()
    flat = st.params.cpu()
    assert len(st.panels) >= 2
    for q in st.panels:
        W = flat[q.src_off:q.src_off + q.K * q.N].reshape(q.K, q.N)
        w16 = W.to(torch.bfloat16).view(torch.int16).numpy()
        if q.gates:
            R = q.N // 4
            ref = np.zeros((R // 4, q.K, 16), np.int16)
            for gate in range(4):
                ref[:, :, gate * 4:gate * 4 + 4] = w16[:, gate * R:(gate + 1) * R].reshape(q.K, R // 4, 4).transpose(1, 0, 2)
        else:
            P = (q.N + 15) // 16
            pad = np.zeros((q.K, P * 16), np.int16)
            pad[:, :q.N] = w16
            ref = pad.reshape(q.K, P, 16).transpose(1, 0, 2)
        assert np.array_equal(pan[q.dst_off:q.dst_off + ref.size], ref.reshape(-1)), (q.src_off, q.K, q.N, q.gates)


@pytest.mark.parametrize("hp_over,B", [({}, 64), (dict(canvas_size=128, max_steps=5, max_digits=4), 32)])
def test_bf16_twins_are_bit_identical_to_fp32_operands(am, hp_over, B):
    """bf16_twins=True (default on the bf16 path): every GEMM / weight-gradient operand is read from a bf16
    twin its producer (or Adam, for the variables) wrote.  The twin is the RNE rounding the fp32-operand
    kernels apply on the way into LDS, so forward outputs, all 36 gradients, both Adam slots, the
    variables and the global norm after several train steps equal the bf16_twins=False run BIT FOR BIT --
    at the bench configuration and at the 128x128 stress configuration (other tiles, several rounds)."""
    hp = dict(HP, **hp_over)
    res = {}
    for tw in (False, True):
        model, *_ = _make(am, B, True, prec="bf16", backward="reference", hp=hp, bf16_twins=tw)
        assert model._twins == tw
        names = {op.kernel for op in model.train_step_ops()}
        assert any("gemm_bf16tw_kernel" in n for n in names) == tw, names
        for _ in range(3):
            model.training()
        torch.cuda.synchronize()
        st = model.store
        res[tw] = dict(params=st.params.clone(), m=st.m.clone(), v=st.v.clone(), grads=st.grads.clone(),
                       gnorm=st.gnorm.clone(), recon=model.reconstruction.clone(), att=model.att.clone(),
                       ml=model.ml.clone(), vrec=model.vrec.clone(), h=model.h.clone())
        if tw:
            # the shadow Adam maintains IS bf16(variables); activation twins are bf16 of their fp32 arrays
            # (the row-major twin of the Wx rows is dropped where the panel twin is its only reader: "exclusive")
            lo = model.store.dims["D"] * 4 * model.store.dims["R"] if st.wx_exclusive else 0
            assert torch.equal(st.params16[lo:].view(torch.bfloat16), st.params[lo:].to(torch.bfloat16))
            _assert_panels_current(st)
            assert torch.equal(model.h16.view(torch.bfloat16), model.h.to(torch.bfloat16))
            assert torch.equal(model.window16.view(torch.bfloat16), model.window.to(torch.bfloat16))
            assert torch.equal(model.d_hid16.view(torch.bfloat16), model.d_hid.to(torch.bfloat16))
            assert torch.equal(model.dgsum16.view(torch.bfloat16), model.dgsum.to(torch.bfloat16))
            assert torch.equal(model.dgates16.view(torch.bfloat16), model.dgates.to(torch.bfloat16))
            assert torch.equal(model.d_genpre16.view(torch.bfloat16), model.d_genpre.to(torch.bfloat16))
    for k in res[False]:
        assert torch.equal(res[False][k], res[True][k]), k


def test_bf16_shadow_follows_host_side_changes_of_the_variables(am):
    """load_state_dict / initialize / sync mark the bf16 shadow stale; the next forward re-derives it (one
    air_bf16_twin launch) also when a captured graph is replayed."""
    model, images, targets, params, noise = _make(am, 16, False, prec="bf16")
    model.use_device_rng()
    model.capture_graph()
    model.forward()
    l0 = float(model.loss)
    p2 = {k: (v * 0.5 if v.ndim == 2 else v) for k, v in params.items()}
    model.load_state_dict(p2)
    assert model.store.shadow_stale
    model.forward()
    torch.cuda.synchronize()
    assert not model.store.shadow_stale
    lo = model.store.dims["D"] * 4 * model.store.dims["R"] if model.store.wx_exclusive else 0
    assert torch.equal(model.store.params16[lo:].view(torch.bfloat16), model.store.params[lo:].to(torch.bfloat16))
    _assert_panels_current(model.store)
    ref, *_ = _make(am, 16, False, prec="bf16", bf16_twins=False)
    ref.use_device_rng()
    ref.load_state_dict(p2)
    ref.forward()
    assert float(model.loss) == float(ref.loss) and float(model.loss) != l0


@pytest.mark.parametrize("prec,twins", [("bf16", True), ("bf16", False), ("fp32", False)])
def test_multi_step_graphs_are_bit_identical_to_eager_steps(am, prec, twins):
    """capture_graph(steps=K > 1), the form bench.py and training.py run: variables, Adam slots, bf16 shadow and
    global_step after 2 replays of 4 steps equal 8 eager steps BIT FOR BIT (noise and schedules are keyed by the
    device-side global_step; ApplyAdam, air_model.py:673-694)."""
    res = {}
    for mode in ("eager", "graph"):
        model, *_ = _make(am, 64, True, prec=prec, backward="reference", bf16_twins=twins)
        model.use_device_rng(seed=11)
        if mode == "graph":
            model.capture_graph(steps=4)
            for _ in range(2):
                model.training()
        else:
            for _ in range(8):
                model.training(eager=True)
        torch.cuda.synchronize()
        st = model.store
        assert int(model.global_step) == 8
        res[mode] = [st.params.clone(), st.m.clone(), st.v.clone(), st.params16.clone(), st.gnorm.clone(), model.scalars.clone()]
        if twins:
            # the row-major shadow follows the variables -- except over Wx when the store keeps only its panel twin
            lo = HP["canvas_size"] ** 2 * 4 * HP["rnn_units"] if st.wx_exclusive else 0
            assert torch.equal(st.params16[lo:].view(torch.bfloat16), st.params[lo:].to(torch.bfloat16))
    for a, b in zip(res["eager"], res["graph"]):
        assert torch.equal(a, b)


def test_state_dict_carries_adam_slots_per_variable(am):
    """state_dict() stores the Adam slots under TensorFlow's slot names (<var>/Adam, <var>/Adam_1), not as the raw flat
    buffers whose offsets depend on the alignment of this build: a dict taken after two steps restores variables AND
    optimizer state (the next step is bit-identical), and flat slots of an unknown layout are refused."""
    model, *_ = _make(am, 16, True)
    for _ in range(2):
        model.training()
    sd = model.state_dict()
    assert "_adam_m" not in sd and "rnn/kernel/Adam" in sd and "vae/rec_mean/biases/Adam_1" in sd
    assert tuple(sd["rnn/kernel/Adam"].shape) == tuple(model.variables["rnn/kernel"].shape)
    assert float(sd["rnn/kernel/Adam_1"].abs().max()) > 0
    model.training()
    torch.cuda.synchronize()
    ref = (model.store.params.clone(), model.store.m.clone(), model.store.v.clone())
    model.store.m.zero_(); model.store.v.zero_()
    model.load_state_dict(sd)
    assert int(model.global_step) == 2
    model.training()
    torch.cuda.synchronize()
    for a, b in zip(ref, (model.store.params, model.store.m, model.store.v)):
        assert torch.equal(a, b)
    # a dict that cannot be loaded leaves the model as it was: validated before anything is written
    before = (model.store.params.clone(), model.store.m.clone(), int(model.global_step))
    old = {**{k: (v * 0 + 3.0 if k != "global_step" else v * 0 + 77) for k, v in sd.items() if "/Adam" not in k},
           "_adam_m": torch.zeros(7), "_adam_v": torch.zeros(7)}
    with pytest.raises(ValueError):
        model.load_state_dict(old)
    half = {k: v for k, v in sd.items() if not k.endswith("rnn/bias/Adam_1")}
    with pytest.raises(ValueError):
        model.load_state_dict(half)
    assert torch.equal(before[0], model.store.params) and torch.equal(before[1], model.store.m)
    assert int(model.global_step) == before[2]
    # ... and the variables alone can still be taken from it (evaluation / demo.py on a checkpoint of an earlier build)
    model.load_state_dict(old, load_optimizer=False)
    assert float(model.variables["rnn/bias"].min()) == 3.0 and int(model.global_step) == 77
    assert torch.equal(before[1], model.store.m)


@pytest.mark.parametrize("train_kw", [dict(prec="fp32"), dict(prec="bf16", bf16_twins=False)])
def test_bf16_shadow_follows_a_train_model_that_does_not_maintain_it(am, train_kw):
    """The bf16 shadow of the variables lives on the shared VariableStore, but only an Adam launch of a model with
    bf16 twins rewrites it.  A train model WITHOUT twins (fp32 precision, bf16_twins=False) changes the
    variables every step: a reuse=True evaluation model WITH twins on the same scope (training.py:95-123 builds such a
    pair) must see the new variables -- its forward equals that of a model without twins, bit for bit, after every
    train step -- instead of running its GEMMs on a stale shadow."""
    tr, images, targets, params, noise = _make(am, 16, True, backward="reference", **train_kw)
    assert not tr._twins
    dev_img, dev_tg = tr.input_images, tr.target_num_digits
    kw = dict(cnn=False, train=False, reuse=True, scope="air", gemm_precision="bf16", **HP)
    ev = am.AIRModel(dev_img, dev_tg, **kw)
    ev_ref = am.AIRModel(dev_img, dev_tg, bf16_twins=False, **kw)
    assert ev._twins and not ev_ref._twins and ev.store is tr.store
    for m in (ev, ev_ref):
        m.set_noise(noise)
        m.set_dynamic(z_pres_prior_log_odds=-2.0)
    losses = []
    for step in range(3):
        ev.forward(); ev_ref.forward()
        torch.cuda.synchronize()
        lo = ev.store.dims["D"] * 4 * ev.store.dims["R"] if ev.store.wx_exclusive else 0
        assert torch.equal(ev.store.params16[lo:].view(torch.bfloat16), ev.store.params[lo:].to(torch.bfloat16)), step
        _assert_panels_current(ev.store)
        assert float(ev.loss) == float(ev_ref.loss), step
        assert torch.equal(ev.reconstruction, ev_ref.reconstruction)
        losses.append(float(ev.loss))
        tr.training()
        assert tr.store.shadow_stale                      # the step changed the variables without touching the shadow
    assert len(set(losses)) == 3


def test_set_backward_switches_the_order_of_a_built_model(am):
    """AIRModel.set_backward (training.py --late-backward): the launch lists are rebuilt for the other order, a captured
    graph is released, variables / Adam slots / global_step carry on; switching there and back gives the same step as a
    model that never switched (same state, same noise key)."""
    model, *_ = _make(am, 16, True, backward="reference")
    model.use_device_rng(seed=5)
    model.capture_graph(steps=2)
    model.training()
    torch.cuda.synchronize()
    assert int(model.global_step) == 2
    kern = lambda m: [op.kernel for op in m.train_step_ops() if "write_bwd" in op.kernel]   # noqa: E731
    assert kern(model) == ["write_bwd_graph_kernel<true>"]
    model.set_backward("reference_carried")
    assert model._graph is None and kern(model) == ["write_bwd_carried_kernel<true>"]
    snap = (model.store.params.clone(), model.store.m.clone(), model.store.v.clone())
    model.set_backward("reference")                        # ... and back: nothing but the launch lists changed
    for a, b in zip(snap, (model.store.params, model.store.m, model.store.v)):
        assert torch.equal(a, b)
    model.training()
    torch.cuda.synchronize()
    ref, *_ = _make(am, 16, True, backward="reference")
    ref.use_device_rng(seed=5)
    for _ in range(3):
        ref.training()
    torch.cuda.synchronize()
    assert int(model.global_step) == int(ref.global_step) == 3
    assert torch.equal(model.store.params, ref.store.params) and torch.equal(model.store.m, ref.store.m)
    model.set_backward("reference_carried")
    model.capture_graph(steps=2)
    model.training()
    torch.cuda.synchronize()
    assert int(model.global_step) == 5 and bool(torch.isfinite(model.store.params).all())
    with pytest.raises(ValueError):
        model.set_backward("sequential")
    for removed in ("taps", "reference_blocked"):                     # measured-worse orders of rounds 1 / 5: gone with ABI 5
        with pytest.raises(ValueError):
            model.set_backward(removed)


@pytest.mark.parametrize("graph_steps", [0, 2])
def test_backward_schedule_switches_at_its_iteration(am, graph_steps):
    """AIRModel(backward=("reference", "reference_carried", N)): the reference's order while global_step < N, the carried
    order from then on; training() makes the switch between two steps (a captured graph is captured again), and the run is
    step for step the one a hand-switched model makes.  A checkpoint load re-reads global_step and picks the order it
    asks for."""
    kern = lambda m: [op.kernel for op in m.train_step_ops() if "write_bwd" in op.kernel]   # noqa: E731
    with pytest.raises(ValueError):
        _make(am, 16, True, backward=("reference", "sequential", 4))
    with pytest.raises(ValueError):
        _make(am, 16, True, backward=("reference", "reference_carried"))
    model, *_ = _make(am, 16, True, backward=("reference", "reference_carried", 4))
    model.use_device_rng(seed=5)
    assert model.backward_schedule == ("reference", "reference_carried", 4) and model.backward == "reference"
    hand, *_ = _make(am, 16, True, backward="reference", scope="hand")
    hand.use_device_rng(seed=5)
    if graph_steps:
        model.capture_graph(steps=graph_steps)
    seen = []
    for it in range(0, 8, max(graph_steps, 1)):
        model.training()
        seen.append(kern(model)[0])
        for _ in range(max(graph_steps, 1)):
            if int(hand.global_step) == 4:
                hand.set_backward("reference_carried")
            hand.training()
        torch.cuda.synchronize()
        assert int(model.global_step) == int(hand.global_step) == it + max(graph_steps, 1)
        assert torch.equal(model.store.params, hand.store.params), it
    n_first = 4 // max(graph_steps, 1)
    assert seen[:n_first] == ["write_bwd_graph_kernel<true>"] * n_first
    assert set(seen[n_first:]) == {"write_bwd_carried_kernel<true>"}
    assert (model._graph is not None) == bool(graph_steps)              # the graph was captured again after the switch
    # back before the switch point: the order follows the loaded global_step
    sd = model.state_dict()
    sd["global_step"] = torch.tensor(1)
    model.load_state_dict(sd)
    model.training()
    assert kern(model) == ["write_bwd_graph_kernel<true>"] and model.backward == "reference"
    # an explicit order ends the schedule
    model.set_backward("reference_carried")
    assert model.backward_schedule is None


def test_load_state_dict_validates_before_it_writes(am):
    model, *_ = _make(am, 8, True)
    model.training()
    torch.cuda.synchronize()
    before = (model.store.params.clone(), model.store.m.clone(), int(model.global_step))
    sd = model.state_dict()
    k = "rnn/kernel"
    bad = dict(sd)
    bad[k] = torch.zeros_like(sd[k]) + 3.0
    bad[k + "/Adam_1"] = torch.zeros(7)                              # a slot of the wrong size
    bad["global_step"] = torch.tensor(99)
    with pytest.raises(ValueError):
        model.load_state_dict(bad)
    assert torch.equal(model.store.params, before[0]) and torch.equal(model.store.m, before[1])
    assert int(model.global_step) == before[2]
    bad = dict(sd)
    bad["global_step"] = "not a number"
    bad[k] = torch.zeros_like(sd[k]) + 3.0
    with pytest.raises(ValueError):
        model.load_state_dict(bad)
    assert torch.equal(model.store.params, before[0])
