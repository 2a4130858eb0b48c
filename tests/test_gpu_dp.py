"""Data-parallel product path on hardware: two rank processes share the one leased MI355X and call
AIRModel.training() -- the `fwd+bwd -> ONE all_reduce of the flat gradient buffer -> grad_sqnorm ->
clip + Adam(prescale 1/world)` branch, eager and as the two captured hipGraphs -- over the gloo
backend (RCCL refuses two ranks on one device; gloo all-reduces device tensors through the host).
Each rank gets half of one B = 2b batch (images and injected noise rows); the result must equal a
single-process step on the whole batch: reference air_model.py:610 (loss = reduce_mean over the
global batch) and :673 (clip_by_global_norm on the AVERAGED gradient)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import air_oracle as ao  # noqa: E402
from oracle.synth import blob_canvases  # noqa: E402

HP = dict(ao.TRAINING_HP)
B2 = 16          # global batch; 8 per rank
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(blank, total=B2):
    images, targets = blob_canvases(total, HP["canvas_size"], HP["max_digits"], seed=23)
    noise = ao.make_noise(HP, total, 5)
    if blank:
        # smooth regime (tests/test_gpu_model.py::_make): no ink -> no log(r + 1e-9) pole under
        # out-of-range sampler residues, z_pres ~ 0.55 -> the canvas stays below 1.  Only here is the
        # fp32 ELBO well-conditioned enough for tight tolerances: with ink, a different GEMM tile
        # shape (M = 3*8 rows per rank vs 3*16) moves it by 1e-3 relative
        images = np.zeros_like(images)
        noise["u"][:] = 0.55
    return images, targets, noise, ao.init_params(HP, 0)


def _run(am, images, targets, noise, params, prec, graph, init_seed, exchange="flat", steps=STEPS):
    am.reset_default_graph()
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False,
                    train=True, scope="air", gemm_precision=prec, seed=init_seed, dp_exchange=exchange, **HP)
    if params is not None:
        m.load_state_dict(params)
    m.set_noise(noise)
    m.set_dynamic(z_pres_prior_log_odds=-2.0)
    if graph:
        m.capture_graph()
    out = []
    for _ in range(steps):
        m.training()
        torch.cuda.synchronize()
        out.append((float(m.loss), float(m.accuracy), float(m.store.gnorm[0])))
    return m, out


def _worker(rank, world, port, prec, graph, blank, q, exchange="flat", steps=STEPS, total=B2, light=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from air import air_model as am
    images, targets, noise, params = _inputs(blank, total)
    b = total // world
    sl = slice(rank * b, (rank + 1) * b)
    # rank 1 deliberately STARTS from different variables (seed) and only rank 0 loads the common
    # ones: sync_parameters() (called by training()/capture_graph()) must make the replicas identical
    m, out = _run(am, images[sl], targets[sl], {k: v[:, sl] for k, v in noise.items()},
                  params if rank == 0 else None, prec, graph, init_seed=100 + rank, exchange=exchange, steps=steps)
    if light:      # eight ranks: a checksum of the raw words instead of three 16 MB arrays per rank through the queue
        words = lambda t: int(t.view(torch.int32).to(torch.int64).sum())      # noqa: E731
        q.put((rank, out, words(m.store.params), int(m.global_step), words(m.store.m),
               m.store.params.cpu().numpy() if rank == 0 else None))
    else:
        q.put((rank, out, m.store.params.cpu().numpy(), int(m.global_step), m.store.m.cpu().numpy(), m.store.grads.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("prec,graph,blank", [("fp32", False, True), ("fp32", True, True), ("fp32", True, False),
                                              ("bf16", True, True), ("bf16", False, False)])
def test_dp2_training_equals_full_batch_step(prec, graph, blank):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, prec, graph, blank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, out, par, step, mom, _ = q.get(timeout=600)
        got[r] = (out, par, step, mom)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # replicas: bit-identical parameters / Adam slots and the same averaged scalars on both ranks
    assert np.array_equal(got[0][1].view(np.int32), got[1][1].view(np.int32))
    assert np.array_equal(got[0][3].view(np.int32), got[1][3].view(np.int32))
    assert got[0][0] == got[1][0]
    assert got[0][2] == got[1][2] == STEPS

    from air import air_model as am
    images, targets, noise, params = _inputs(blank)
    ref, ref_out = _run(am, images, targets, noise, params, prec, False, init_seed=0)
    p_ref = ref.store.params.cpu().numpy()
    p0 = np.zeros_like(p_ref)
    # the step-0 parameters in the flat layout
    am.reset_default_graph()
    tmp = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False,
                      train=True, scope="air", gemm_precision=prec, **HP)
    tmp.load_state_dict(params)
    p0 = tmp.store.params.cpu().numpy()
    report = []
    for i, ((l, a, g), (lr_, ar_, gr_)) in enumerate(zip(got[0][0], ref_out)):
        report.append((abs(l - lr_) / abs(lr_), abs(g - gr_) / gr_))
        if blank or i == 0:
            assert abs(a - ar_) < 1e-6
        if blank:
            tol = 2e-5 if prec == "fp32" else 2e-3    # bf16: dW contracts over bf16-rounded rows in another order
            assert abs(l - lr_) / abs(lr_) < tol, (l, lr_)
            assert abs(g - gr_) / gr_ < 10 * tol, (g, gr_)           # norm of the AVERAGED gradient
        elif i == 0:
            # with ink only the FIRST step is comparable: its gradient carries the reference's rounding
            # residue (|g| ~1e3 x the exact one, chaotic in the last bit of every input), so the
            # parameters after one update differ between any two evaluation orders
            assert abs(l - lr_) / abs(lr_) < 1e-2, (l, lr_)          # the repo-wide ELBO tolerance with ink
    d_dp, d_ref = got[0][1] - p0, p_ref - p0
    assert np.linalg.norm(d_ref) > 0
    # Adam's first steps move every weight by ~lr * sign(g): elements with |g| near 0 may flip, so
    # the update is compared as a whole (relative L2)
    rel = np.linalg.norm(d_dp - d_ref) / np.linalg.norm(d_ref)
    print("dp2 %s graph=%s blank=%s: (loss, gnorm) rel per step %r, update rel-L2 %.3e" % (prec, graph, blank, report, rel))
    if blank:
        assert rel < (5e-3 if prec == "fp32" else 1e-1), rel


def _dp2(prec, graph, blank, exchange, steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, prec, graph, blank, q, exchange, steps)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, out, par, step, mom, grads = q.get(timeout=600)
        got[r] = dict(out=out, params=par, step=step, m=mom, grads=grads)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("prec,graph", [("fp32", True), ("bf16", True), ("bf16", False)])
def test_dp2_factor_exchange_equals_flat_all_reduce(prec, graph):
    """dp_exchange="factors": the LSTM input-weight gradient dWx = X^T.(sum_t dgates) (64 % of the gradient
    elements) is never all-reduced -- X and sum_t dgates (rank <= B per GPU) are all-gathered and every rank
    contracts the gathered 2b rows itself; the rest of the flat buffer is all-reduced.  Against the flat all-reduce
    on the same two ranks: every other gradient BIT-IDENTICAL, dWx and the lstm bias equal up to the fp32 order of
    the contraction, replicas bit-identical, and the same variables after three steps in the smooth regime
    (reference semantics kept: air_model.py:610 mean over the global batch, :673 clip on the averaged gradient)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    D4R = HP["canvas_size"] ** 2 * 4 * HP["rnn_units"]
    one = {ex: _dp2(prec, graph, False, ex, 1) for ex in ("flat", "factors")}
    for ex in one:
        assert np.array_equal(one[ex][0]["params"].view(np.int32), one[ex][1]["params"].view(np.int32))   # replicas
        assert np.array_equal(one[ex][0]["grads"].view(np.int32), one[ex][1]["grads"].view(np.int32))
    gf, gx = one["flat"][0]["grads"], one["factors"][0]["grads"]
    nb = 4 * HP["rnn_units"]
    off_bias = D4R + HP["rnn_units"] * nb                       # lstm_kernel = [Wx rows | Wh rows], then lstm_bias
    assert np.array_equal(gf[D4R:off_bias], gx[D4R:off_bias])   # Wh
    assert np.array_equal(gf[off_bias + nb:], gx[off_bias + nb:])   # everything after the LSTM bias, incl. loss/accuracy
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64))  # noqa: E731
    assert np.abs(gf[:D4R]).max() > 0
    assert rel(gx[:D4R], gf[:D4R]) < 2e-6, rel(gx[:D4R], gf[:D4R])
    assert rel(gx[off_bias:off_bias + nb], gf[off_bias:off_bias + nb]) < 2e-6
    assert one["flat"][0]["out"][0][2] == pytest.approx(one["factors"][0]["out"][0][2], rel=1e-5)        # global norm
    three = {ex: _dp2(prec, graph, True, ex, STEPS) for ex in ("flat", "factors")}
    assert np.array_equal(three["factors"][0]["params"].view(np.int32), three["factors"][1]["params"].view(np.int32))
    assert three["factors"][0]["step"] == STEPS
    assert rel(three["factors"][0]["params"], three["flat"][0]["params"]) < 1e-5


@pytest.mark.parametrize("exchange,graph", [("flat", False), ("flat", True), ("factors", False), ("factors", True)])
def test_dp8_configs2_world_size_equals_one_b512_step(exchange, graph):
    """BASELINE configs[2] at its REAL world size in the only form a 1-GPU lease allows: eight rank processes x b = 64
    (global batch 512) share the one MI355X over gloo; both gradient exchanges, eager and as [fwd+bwd graph] | collective |
    [clip+Adam graph].  Against ONE b = 512 step of a single process on blank canvases (smooth regime): loss 2e-5,
    norm of the AVERAGED gradient 2e-4 (reference air_model.py:610-611: reduce_mean over the global batch; :673: clip after
    the reduction), accuracy equal, replicas bit-identical after three steps, the 1/8 scaling of loss / accuracy that rode
    in the all-reduce as sums.  The factor exchange contracts K = 8 x 64 = 512 gathered rows per rank.
    Not a scaling measurement -- RCCL over xGMI at 8 ranks has not run anywhere (no multi-GPU lease)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, total = 8, 512
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "fp32", graph, True, q, exchange, STEPS, total, True))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, out, psum, step, msum, par = q.get(timeout=900)
        got[r] = dict(out=out, psum=psum, step=step, msum=msum, params=par)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    for r in range(1, world):
        assert got[r]["psum"] == got[0]["psum"] and got[r]["msum"] == got[0]["msum"], r      # replicas bit-identical
        assert got[r]["out"] == got[0]["out"], r                                              # the same averaged scalars everywhere
        assert got[r]["step"] == STEPS
    from air import air_model as am
    images, targets, noise, params = _inputs(True, total)
    ref, ref_out = _run(am, images, targets, noise, params, "fp32", False, init_seed=0)
    rep = []
    for (l, a, g), (lr_, ar_, gr_) in zip(got[0]["out"], ref_out):
        rep.append((abs(l - lr_) / abs(lr_), abs(g - gr_) / gr_))
        assert abs(a - ar_) < 1e-6                               # accuracy: a mean of 512 indicator values, /8 of the rank sums
        assert abs(l - lr_) / abs(lr_) < 2e-5, (l, lr_)
        assert abs(g - gr_) / gr_ < 2e-4, (g, gr_)
    am.reset_default_graph()
    tmp = am.AIRModel(torch.tensor(images[:8], device="cuda"), torch.tensor(targets[:8], device="cuda"), cnn=False,
                      train=True, scope="air", gemm_precision="fp32", **HP)
    tmp.load_state_dict(params)
    p0 = tmp.store.params.cpu().numpy()
    d_dp, d_ref = got[0]["params"] - p0, ref.store.params.cpu().numpy() - p0
    rel = np.linalg.norm(d_dp - d_ref) / np.linalg.norm(d_ref)
    print("dp8 %s graph=%s: (loss, gnorm) rel per step %r, update rel-L2 %.3e" % (exchange, graph, rep, rel))
    assert rel < 5e-3, rel


def test_bench_eight_rank_flow_on_one_device():
    """`python bench.py --gpus 8` (configs[2]: 8 x b = 64) end to end on the one leased GPU over gloo: the line is labelled
    as a flow test, carries global_batch 512, bit-identical replicas and both exchanges' step times."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AIR_BENCH_SAME_DEVICE="1", AIR_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "10", "--warmup", "2"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 512 and d["config"]["parallelism"] == "dp8"
    assert d["replicas_bit_identical"] is True and "NOT a scaling measurement" in d["test_mode"]
    assert set(d["distributed"]["ms_per_step_by_exchange"]) == {"flat", "factors"}
    out = os.path.join(root, "gpurun_out", "r05")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_dp8_same_device_gloo.json"), "w") as f:
        f.write(lines[0] + "\n")


def test_bench_multi_rank_flow_on_one_device(tmp_path):
    """`python bench.py --gpus 2` end to end -- rank spawning, parameter sync, the [fwd+bwd graph] ->
    all_reduce -> [clip+Adam graph] loop, max-over-ranks timing, the collective's own timing, ONE JSON
    line from rank 0 -- with both ranks on the one leased GPU over gloo (a flow test: the line is
    labelled as not a scaling measurement).  The same command without the test hooks must refuse to
    run on a 1-GPU box."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AIR_BENCH_SAME_DEVICE="1", AIR_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "4"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert d["replicas_bit_identical"] is True
    assert d["value"] > 0 and abs(d["value"] - 128 * 12 / (d["ms_per_step"] * 12 * 1e-3)) / d["value"] < 1e-3
    assert d["allreduce"]["bytes"] >= 4 * 4011643 and d["allreduce"]["bytes"] % 16 == 0 and d["allreduce"]["us"] > 0
    assert "NOT a scaling measurement" in d["test_mode"]
    if torch.cuda.device_count() < 2:
        q = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "0"], cwd=root,
                           capture_output=True, text=True, timeout=300)
        assert q.returncode != 0 and "refusing" in q.stderr and not any(l.startswith("{") for l in q.stdout.splitlines())


_RCCL_ONE_RANK = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tf-attend-infer-repeat_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from oracle import air_oracle as ao
from oracle.synth import blob_canvases
from air import air_model as am
HP = dict(ao.TRAINING_HP)
images, targets = blob_canvases(8, HP["canvas_size"], HP["max_digits"], seed=23)
for exchange in ("flat", "factors"):
    am.reset_default_graph()
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                    scope="air", gemm_precision="bf16", seed=0, dp_exchange=exchange, **HP)
    m.sync_parameters()                                    # broadcast over RCCL
    s = m._stream()
    m._run_forward(s, finalize=False)
    m._run_backward(s, for_update=False, fused_finalize=True)
    torch.cuda.synchronize()
    before = m.store.grads.clone()
    m._dp_exchange_gradients()                             # all_reduce (+ all_gather_into_tensor) over RCCL
    torch.cuda.synchronize()
    assert torch.equal(before, m.store.grads), exchange    # one rank: the sum over ranks is the identity
    if exchange == "factors":
        f = m._dp_factors()
        assert torch.equal(f["x_all"], m.input_images) and torch.equal(f["dg_all"], m.dgsum)
    assert float(before.abs().max()) > 0
    # and a whole train step on the single-rank process group (world == 1: the fused single-GPU branch)
    m.training()
    torch.cuda.synchronize()
print("RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()))
dist.destroy_process_group()
'''


def test_rccl_collectives_execute_on_the_product_buffers(tmp_path):
    """The `nccl` (= RCCL) calls of the data-parallel branch -- broadcast of the variables, ONE all_reduce of the flat
    gradient buffer, all_gather_into_tensor of the dWx factors + the all_reduce of the tail -- executed on this box's one
    GPU with a single-rank process group (RCCL refuses two ranks per device, and one rank is all a 1-GPU lease allows):
    not a scaling test, but the calls, dtypes, sizes and alignments of the buffers the product hands to RCCL are
    exercised on hardware, and a one-rank sum must leave them unchanged."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), root, str(_free_port())], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RCCL" in r.stdout


_RCCL_IN_GRAPH = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tf-attend-infer-repeat_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                  AIR_DP_FORCE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from oracle import air_oracle as ao
from oracle.synth import blob_canvases
from air import air_model as am
HP = dict(ao.TRAINING_HP)
images, targets = blob_canvases(64, HP["canvas_size"], HP["max_digits"], seed=23)

def run(exchange, mode, prec):
    am.reset_default_graph()
    m = am.AIRModel(torch.tensor(images, device="cuda"), torch.tensor(targets, device="cuda"), cnn=False, train=True,
                    scope="air", gemm_precision=prec, seed=0, noise_seed=11, dp_exchange=exchange,
                    annealing_schedules=ao.TRAINING_ANNEALING, **HP)
    assert m._dp() and m._collectives_capturable()
    assert [op.kernel for op in m._optimizer_ops()][0] == "grad_sqnorm_kernel"     # the DP optimizer: norm pass after the exchange
    if mode == "graph":
        m.capture_graph(steps=4)
        assert m._graph[1] is None and m._graph_steps == 4          # ONE graph: the collective is inside
        for _ in range(2):
            m.training()
    elif mode == "two_graphs":
        os.environ["AIR_DP_GRAPH_COLLECTIVE"] = "0"
        try:
            m.capture_graph()
        finally:
            del os.environ["AIR_DP_GRAPH_COLLECTIVE"]
        assert m._graph[1] is not None
        for _ in range(8):
            m.training()
    else:
        for _ in range(8):
            m.training(eager=True)
    torch.cuda.synchronize()
    st = m.store
    assert int(m.global_step) == 8
    return [st.params.clone(), st.m.clone(), st.v.clone(), st.grads.clone(), st.gnorm.clone(), m.scalars.clone()]

for prec in ("bf16", "fp32"):
    for exchange in ("flat", "factors"):
        ref = run(exchange, "eager", prec)
        assert float(ref[0].abs().max()) > 0 and np.isfinite(float(ref[4]))
        for mode in ("graph", "two_graphs"):
            got = run(exchange, mode, prec)
            for a, b in zip(ref, got):
                assert torch.equal(a, b), (prec, exchange, mode)
    print("in-graph", prec, "ok")
print("RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()))
dist.destroy_process_group()
'''


def test_rccl_collective_captured_inside_the_train_step_graph(tmp_path):
    """Data parallel over RCCL, the protocol of the 1-GPU number: capture_graph(steps=k) records
    ([fwd+bwd] -> gradient exchange -> [grad_sqnorm + clip + Adam]) x k into ONE hipGraph -- the collective is stream
    work, torch.cuda.graph captures it -- instead of one step per replay with a host-enqueued collective between two
    graphs.  On this box's single GPU the process group has one rank (AIR_DP_FORCE=1 makes the model run the DP branch
    on it): 2 replays of 4 steps == 8 eager DP steps == 8 steps of the two-graph form BIT FOR BIT (variables, Adam
    slots, gradients, global norm, loss / accuracy), flat and factor exchange, bf16 and fp32
    (reference semantics: air_model.py:610 reduce_mean over the global batch, :673 clip on the averaged gradient)."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_in_graph.py"
    script.write_text(_RCCL_IN_GRAPH)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AIR_DP_GRAPH_COLLECTIVE", None)
    r = subprocess.run([sys.executable, str(script), root, str(_free_port())], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "in-graph bf16 ok" in r.stdout and "in-graph fp32 ok" in r.stdout and "RCCL" in r.stdout
