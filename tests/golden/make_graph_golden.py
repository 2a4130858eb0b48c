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


if __name__ == "__main__":
    if "--compose-only" in sys.argv:
        compose()
    else:
        main()
