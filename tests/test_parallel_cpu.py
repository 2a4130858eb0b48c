"""Data-parallel path on CPU (gloo, world_size 2): the product's flat gradient buffer
(VariableStore layout + TF-named views + loss/accuracy tail) goes through ONE all_reduce,
then clip-by-global-norm on the AVERAGED gradient and TF-Adam must equal the single-process
update on the full batch (SURVEY 5.8 / appendix D "DP equivalence").  Gradients come from the
fp64 oracle twin; no GPU kernels are called."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import air_oracle as ao
from oracle import air_oracle_torch as at
from oracle.synth import blob_canvases

HP = dict(ao.TRAINING_HP)
B = 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads(images, targets, noise, params):
    f64 = torch.float64
    pt = at.to_torch(params, dtype=f64, requires_grad=True)
    out, g = at.loss_and_grads(pt, torch.tensor(images, dtype=f64), torch.tensor(targets),
                               at.to_torch(noise, dtype=f64), HP, -2.0)
    return out, g


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from air.air_model import VariableStore
    images, targets = blob_canvases(B, 50, 2, seed=17)
    params = ao.init_params(HP, 0)
    noise = ao.make_noise(HP, B, 1)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    out, g = _grads(images[sl], targets[sl], {k: v[:, sl] for k, v in noise.items()}, params)
    hp = {k: HP[k] for k in ("canvas_size", "windows_size", "rnn_units", "vae_latent_dimensions",
                             "vae_recognition_units", "vae_generative_units", "scale_hidden_units",
                             "shift_hidden_units", "z_pres_hidden_units")}
    store = VariableStore(hp, torch.device("cpu"))
    for k, v in g.items():
        store.gradients[k].copy_(v.float())
    store.grads[store.n] = float(out["loss"])
    store.grads[store.n + 1] = float(out["accuracy"])
    dist.all_reduce(store.grads)                       # the ONE collective of the step
    # clip AFTER the all-reduce, on the averaged gradient (air_model.py:673)
    avg = {k: store.gradients[k].double().clone() / world for k in g}
    if rank == 0:
        q.put(({k: v.numpy() for k, v in avg.items()}, float(store.grads[store.n]) / world,
               int(store.n), int(store.num_trainable)))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4])
def test_dp_allreduce_equals_full_batch(world):
    """world 2 and world 4 (one image per rank): sum over the ranks / world == the full-batch mean"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    avg, loss_avg, n, ntrain = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ntrain == 4011643 and n >= ntrain and n % 4 == 0
    images, targets = blob_canvases(B, 50, 2, seed=17)
    params = ao.init_params(HP, 0)
    noise = ao.make_noise(HP, B, 1)
    out, g = _grads(images, targets, noise, params)
    assert abs(loss_avg - float(out["loss"])) / abs(float(out["loss"])) < 1e-5
    for k, v in g.items():
        ref = v.double().numpy()
        err = np.linalg.norm(avg[k] - ref) / max(np.linalg.norm(ref), 1e-12)
        assert err < 1e-5, (k, err)          # fp32 transport of fp64 grads
    # clip on the averaged gradient == clip of the full-batch gradient
    gn_avg = np.sqrt(sum((a ** 2).sum() for a in avg.values()))
    gn_ref = np.sqrt(sum((v.double().numpy() ** 2).sum() for v in g.values()))
    assert abs(gn_avg - gn_ref) / gn_ref < 1e-5


def test_variable_store_views_cover_tf_names():
    from air.air_model import VariableStore
    hp = {k: HP[k] for k in ("canvas_size", "windows_size", "rnn_units", "vae_latent_dimensions",
                             "vae_recognition_units", "vae_generative_units", "scale_hidden_units",
                             "shift_hidden_units", "z_pres_hidden_units")}
    st = VariableStore(hp, torch.device("cpu"))
    shapes = ao.param_shapes(HP)
    assert list(st.variables.keys()) == list(shapes.keys())       # air-model.index names
    for k, shp in shapes.items():
        assert tuple(st.variables[k].shape) == tuple(shp), k
    # views alias the flat buffer: writing a TF-named view is visible in the flat storage
    st.params.zero_()
    st.variables["shift/mean/output/weights"].fill_(2.0)
    assert float(st.params.sum()) == 2.0 * 64 * 2
    st.variables["vae/rec_log_variance/weights"].fill_(1.0)
    assert float(st.P["ml_w"][:, 50:].sum()) == 256 * 50 and float(st.P["ml_w"][:, :50].sum()) == 0
    # state_dict round trip keeps TF names
    sd = st.state_dict()
    st.params.zero_()
    st.load_state_dict(sd)
    assert float(st.variables["shift/mean/output/weights"].sum()) == 2.0 * 128
