"""Generates tests/golden/graph_b64.npz by EXECUTING the reference's own serialized training graph
(/root/reference/model/air-model.meta, written by TensorFlow 1.3 for the 270k-iteration run) with
the numpy dataflow executor oracle/graphdef_exec.py.  Run from the repo root, in the build
container (the reference does not travel to the GPU box; this fixture does):

    python tests/golden/make_graph_golden.py

Unlike tests/golden/air_b4.npz (a snapshot of the oracle restatement), these vectors come from the
reference's graph itself: forward outputs, the gradients `tf.gradients` built (evaluated in fp64 =
the graph's exact math, and in fp32 = with the rounding residue of the reference's op order), the
clip + ApplyAdam update, and the tensors at the interface of the two sampler backward kernels.
Inputs are regenerated from seeds by the tests (np.random.RandomState is platform-stable).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import air_oracle as ao  # noqa: E402
from oracle import graphdef_exec as gx  # noqa: E402
from oracle.synth import blob_canvases  # noqa: E402

META = "/root/reference/model/air-model.meta"
HP = dict(ao.TRAINING_HP)
B = 64                  # the train model's batch is static in the saved graph (training.py:30)
KB = 16                 # images kept for the kernel-level vectors
SUB = 2048              # gradient elements kept per large tensor
SEEDS = dict(images=3, params=0, noise=1)

FWD_KEYS = ("loss", "accuracy", "reconstruction", "reconstruction_loss", "rec_num_digits", "rec_scales",
            "rec_shifts", "rec_st_back", "z_pres_probs", "z_pres_kls", "scale_kls", "shift_kls", "vae_kls",
            "z_pres_prior_log_odds")


def subsample_index(name, numel):
    """the fixed element subset of a tensor stored in the fixture (all of it when small)"""
    if numel <= 2 * SUB:
        return np.arange(numel)
    rng = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
    return np.sort(rng.choice(numel, SUB, replace=False))


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
    return h


def inputs(batch=B, seed_images=SEEDS["images"], seed_noise=SEEDS["noise"]):
    images, targets = blob_canvases(batch, HP["canvas_size"], HP["max_digits"], seed=seed_images)
    return images, targets, ao.init_params(HP, SEEDS["params"]), ao.make_noise(HP, batch, seed_noise)


def main():
    version, nodes = gx.load_graph(META)
    assert version == "1.3.0"
    out = {}
    images, targets, params, noise = inputs()
    T = gx.output_tensors("air")
    adam = gx.adam_nodes(nodes)
    names = list(ao.param_shapes(HP).keys())
    assert sorted(adam) == sorted(names)

    # ---- train model, global_step 0, fp32: forward + the residue-carrying fp32 backward
    ex = gx.Executor(nodes, gx.air_feeds(nodes, params, images, targets, noise, 0), np.float32)
    vals = dict(zip(FWD_KEYS, ex.run([T[k] for k in FWD_KEYS])))
    for k in FWD_KEYS:
        out["train0/" + k] = np.asarray(vals[k])
    out["train0/rec_windows_first"] = np.asarray(ex.run([T["rec_windows"]])[0])[:KB]
    out["train0/latent_samples_first"] = np.asarray(ex.run([T["_graph_latent_samples"]])[0])[:KB]
    trips = ex.trip_count(gx.FWD_FRAME)
    out["train0/steps_executed"] = np.int32(trips)
    g32 = ex.run([gx.raw_gradient_tensor(nodes, adam[k]) for k in names])
    out["train0/global_norm_fp32"] = np.float32(ex.run(["air/training/global_norm/global_norm"])[0])
    for k, g in zip(names, g32):
        out["train0/grad32_norm/" + k] = np.float32(np.linalg.norm(g.astype(np.float64)))
    assert ex.trip_count(gx.BWD_FRAME) == trips
    # sampler-kernel interface tensors, per forward step t (backward iteration trips-1-t), first KB images
    for t in range(trips):
        j = trips - 1 - t
        fw = ex.run(list(gx.SAMPLER_FWD_TENSORS.values()), {gx.FWD_FRAME: t})
        for k, v in zip(gx.SAMPLER_FWD_TENSORS, fw):
            out["kern/t%d/%s" % (t, k)] = np.asarray(v)[:KB]
        bw = ex.run(list(gx.SAMPLER_BWD_TENSORS.values()), {gx.BWD_FRAME: j})
        for k, v in zip(gx.SAMPLER_BWD_TENSORS, bw):
            out["kern/t%d/%s" % (t, k)] = np.asarray(v)[:KB].reshape(KB, -1).squeeze()

    # ---- the same step in fp64: the graph's exact math (gradients, global norm, Adam update)
    ex64 = gx.Executor(nodes, gx.air_feeds(nodes, params, images, targets, noise, 0, float_dtype=np.float64), np.float64)
    out["train0/loss_fp64"] = np.float64(ex64.run([T["loss"]])[0])
    g64 = ex64.run([gx.raw_gradient_tensor(nodes, adam[k]) for k in names])
    out["train0/global_norm_fp64"] = np.float64(ex64.run(["air/training/global_norm/global_norm"])[0])
    ex64.run([adam[k].name for k in names])
    for k, g in zip(names, g64):
        idx = subsample_index(k, g.size)
        out["train0/grad64_norm/" + k] = np.float64(np.linalg.norm(g))
        out["train0/grad64_sub/" + k] = g.reshape(-1)[idx]
        a = ex64.assigned[gx.SCOPE + k]
        out["train0/adam64_delta_sub/" + k] = (a["var"] - np.asarray(params[k], np.float64)).reshape(-1)[idx]

    # ---- train model at global_step 3000 (annealed prior log-odds log(10 + 1e-9)), other images / noise
    images2, targets2, _, noise2 = inputs(seed_images=11, seed_noise=7)
    ex2 = gx.Executor(nodes, gx.air_feeds(nodes, params, images2, targets2, noise2, 3000), np.float32)
    keys2 = ("loss", "accuracy", "reconstruction_loss", "rec_num_digits", "z_pres_kls", "vae_kls", "z_pres_prior_log_odds")
    for k, v in zip(keys2, ex2.run([T[k] for k in keys2])):
        out["train3000/" + k] = np.asarray(v)
    out["train3000/steps_executed"] = np.int32(ex2.trip_count(gx.FWD_FRAME))

    # ---- test model `air_1` (train=False: z_pres rounded, dynamic batch), B = 4, global_step 40000
    images4, targets4, _, noise4 = inputs(batch=4)
    T1 = gx.output_tensors("air_1")
    ex4 = gx.Executor(nodes, gx.test_model_feeds(params, images4, targets4, noise4, 40000), np.float32)
    keys4 = FWD_KEYS + ("rec_windows", "_graph_latent_samples")
    for k, v in zip(keys4, ex4.run([T1[k] for k in keys4])):
        out["test40000_b4/" + k] = np.asarray(v)
    out["test40000_b4/steps_executed"] = np.int32(ex4.trip_count("air_1/rnn/while/air_1/rnn/while/"))

    path = os.path.join(ROOT, "tests", "golden", "graph_b64.npz")
    np.savez_compressed(path, **out)
    print("wrote %d arrays, %.0f KB; loss %.4f (fp64 %.4f), |g| fp32 %.4e fp64 %.4e, T' = %d" %
          (len(out), os.path.getsize(path) / 1024, out["train0/loss"], out["train0/loss_fp64"],
           out["train0/global_norm_fp32"], out["train0/global_norm_fp64"], trips))
    compose(nodes)
    carried(nodes)
    chain(nodes)


# tensors at the interface of the compose kernel (air_write_fwd: the write transformer, the masked canvas accumulation,
# the VAE KL, the running loss and the Bernoulli cross-entropy; air_model.py:351-366, 429-439, 479-496, 580-593) that
# graph_b64.npz does not already hold: the posterior mean / log-variance of every step and the loop's exit values
COMPOSE_FWD_TENSORS = {"rec_mean": gx.W + "vae/rec_mean/BiasAdd", "rec_log_variance": gx.W + "vae/rec_log_variance/BiasAdd"}
COMPOSE_EXIT_TENSORS = {"running_loss": "air/rnn/while/Exit_5", "loss_per_item": "air/add_1",
                        "running_recon": "air/rnn/while/Exit_4"}


def compose(nodes=None):
    """tests/golden/graph_b64_compose.npz: the train0 run of main() again (same seeds), first KB images"""
    if nodes is None:
        _, nodes = gx.load_graph(META)
    images, targets, params, noise = inputs()
    ex = gx.Executor(nodes, gx.air_feeds(nodes, params, images, targets, noise, 0), np.float32)
    out = {}
    for k, v in zip(COMPOSE_EXIT_TENSORS, ex.run(list(COMPOSE_EXIT_TENSORS.values()))):
        out["train0/" + k] = np.asarray(v)[:KB]
    trips = ex.trip_count(gx.FWD_FRAME)
    for t in range(trips):
        for k, v in zip(COMPOSE_FWD_TENSORS, ex.run(list(COMPOSE_FWD_TENSORS.values()), {gx.FWD_FRAME: t})):
            out["kern/t%d/%s" % (t, k)] = np.asarray(v)[:KB]
    path = os.path.join(ROOT, "tests", "golden", "graph_b64_compose.npz")
    np.savez_compressed(path, **out)
    print("wrote %d arrays, %.0f KB -> %s" % (len(out), os.path.getsize(path) / 1024, path))


def carried(nodes=None, order="carried16"):
    """tests/golden/graph_b64_carried.npz: the fp32 backward of the train0 run once more with the graph's ONE
    UnsortedSegmentSum evaluated in the carried16 order (oracle.carried_segment_sum; AIRModel(backward="reference_carried"))
    -- per-variable gradient norms and the global norm, next to graph_b64.npz's grad32_norm/* (the sequential order) and
    grad64_norm/* (exact)."""
    if nodes is None:
        _, nodes = gx.load_graph(META)
    images, targets, params, noise = inputs()
    adam = gx.adam_nodes(nodes)
    names = list(ao.param_shapes(HP).keys())
    out = {}
    gx.SEGMENT_SUM_ORDER = order
    try:
        ex = gx.Executor(nodes, gx.air_feeds(nodes, params, images, targets, noise, 0), np.float32)
        g32 = ex.run([gx.raw_gradient_tensor(nodes, adam[k]) for k in names])
        out["train0/global_norm_fp32"] = np.float32(ex.run(["air/training/global_norm/global_norm"])[0])
        for k, g in zip(names, g32):
            out["train0/grad32_norm/" + k] = np.float32(np.linalg.norm(g.astype(np.float64)))
        trips = ex.trip_count(gx.FWD_FRAME)
        for t in range(trips):
            bw = ex.run([gx.SAMPLER_BWD_TENSORS["d_gen_pre"]], {gx.BWD_FRAME: trips - 1 - t})
            out["kern/t%d/d_gen_pre" % t] = np.asarray(bw[0])[:KB].reshape(KB, -1)
    finally:
        gx.SEGMENT_SUM_ORDER = "sequential"
    path = os.path.join(ROOT, "tests", "golden", "graph_b64_%s.npz" % order[:-2])
    np.savez_compressed(path, **out)
    print("wrote %d arrays, %.0f KB -> %s; |g| fp32 %s %.4e" % (len(out), os.path.getsize(path) / 1024, path, order,
                                                               out["train0/global_norm_fp32"]))


# tensors at the remaining kernel interfaces of the backward of step t (the VAE data-gradient chain vae.py:10-43
# backwards, the heads' hidden layers, the BasicLSTMCell backward air_model.py:286) -- with write_bwd / attend_bwd /
# compose already pinned, every kernel of the default backward is then fed the executed graph's own tensors
V, GV, GR = gx.W + "vae/", gx.G + gx.W + "vae/", gx.G + gx.W + "rnn/"
HEADS = ("scale/mean", "scale/log_variance", "shift/mean", "shift/log_variance", "z_pres/log_odds")   # column order of whid / d_hid
CHAIN_FWD_TENSORS = {
    "gen_act1": V + "generative_1/generative_1/Softplus", "gen_act2": V + "generative_2/generative_2/Softplus",
    "rec_act1": V + "recognition_1/recognition_1/Softplus", "rec_act2": V + "recognition_2/recognition_2/Softplus",
    "lstm_i": gx.W + "rnn/Sigmoid_1", "lstm_j": gx.W + "rnn/Tanh", "lstm_f": gx.W + "rnn/Sigmoid", "lstm_o": gx.W + "rnn/Sigmoid_2",
    "c_prev": gx.W + "Identity_2", "c_new": gx.W + "rnn/add_1",
}
CHAIN_BWD_TENSORS = {
    "d_gen2": GV + "generative_2/generative_2/Softplus_grad/SoftplusGrad",      # [B,512] wrt generative_2's pre-activation
    "d_gen1": GV + "generative_1/generative_1/Softplus_grad/SoftplusGrad",      # [B,256]
    "d_z": GV + "generative_1/MatMul_grad/MatMul",                              # [B,Z] wrt the latent sample
    "d_mean": gx.G + "AddN_13", "d_log_variance": gx.G + "AddN_14",             # [B,Z] each: reparameterisation + KL
    "d_rec2": GV + "recognition_2/recognition_2/Softplus_grad/SoftplusGrad",    # [B,256]
    "d_rec1": GV + "recognition_1/recognition_1/Softplus_grad/SoftplusGrad",    # [B,512]
    "d_window_vae": GV + "recognition_1/MatMul_grad/MatMul",                    # [B,784] the VAE's share of d loss / d glimpse
    "dh_total": gx.G + "AddN_33",                                               # [B,R] d loss / d h'[t]: heads + recurrence
    "dh_rec": gx.G + gx.W + "Merge_3_grad/tuple/control_dependency_1",          # [B,R] the recurrence's share (from step t+1)
    "dgates": GR + "split_grad/concat",                                         # [B,4R] i, j, f, o
    "dc_prev": GR + "mul_grad/tuple/control_dependency",                        # [B,R] d loss / d c[t-1]
    "dc_in": GR + "add_1_grad/tuple/control_dependency",                        # [B,R] d loss / d c[t] as it reaches add_1
}
for _h in HEADS:
    CHAIN_BWD_TENSORS["d_hid/" + _h] = gx.G + gx.W + _h + "/hidden/hidden/Relu_grad/ReluGrad"    # [B,64] wrt the pre-activation


def chain(nodes=None):
    """tests/golden/graph_b64_chain.npz: the train0 run of main() again (same seeds, fp32), first KB_CHAIN images"""
    KBC = 8
    if nodes is None:
        _, nodes = gx.load_graph(META)
    images, targets, params, noise = inputs()
    ex = gx.Executor(nodes, gx.air_feeds(nodes, params, images, targets, noise, 0), np.float32)
    out = {}
    trips = ex.trip_count(gx.FWD_FRAME)
    for t in range(trips):
        for k, v in zip(CHAIN_FWD_TENSORS, ex.run(list(CHAIN_FWD_TENSORS.values()), {gx.FWD_FRAME: t})):
            out["kern/t%d/%s" % (t, k)] = np.asarray(v)[:KBC]
        for k, v in zip(CHAIN_BWD_TENSORS, ex.run(list(CHAIN_BWD_TENSORS.values()), {gx.BWD_FRAME: trips - 1 - t})):
            out["kern/t%d/%s" % (t, k)] = np.asarray(v)[:KBC]
    path = os.path.join(ROOT, "tests", "golden", "graph_b64_chain.npz")
    np.savez_compressed(path, **out)
    print("wrote %d arrays, %.0f KB -> %s" % (len(out), os.path.getsize(path) / 1024, path))
    for k in sorted(out):
        if k.startswith("kern/t0/"):
            print("  %-40s %-12s max|.| %.3e" % (k, out[k].shape, np.abs(out[k]).max()))


if __name__ == "__main__":
    if "--compose-only" in sys.argv:
        compose()
    elif "--chain-only" in sys.argv:
        chain()
    elif "--carried-only" in sys.argv:
        carried()
    else:
        main()
