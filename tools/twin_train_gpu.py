"""Diagnostic (not the product path): trains the torch-autograd twin of the reference
(oracle/air_oracle_torch.py: un-fused fp32 ops, autograd's own residue-carrying gradients) with
torch-ROCm kernels on the same data, annealing and optimizer as training.py -- an independent
implementation of "the reference's fp32 autodiff" to compare success rates with.

  python tools/twin_train_gpu.py <seed> <iterations> [<backgrounds.npz>:<key> [<max intensity>]] [--no-graph] [--deadline-min M]

One train iteration (noise, forward, autograd backward, global-norm clip, TF-style Adam: ~1 700 small launches) is recorded
ONCE as a hipGraph and replayed -- the un-fused op sequence is unchanged, only its launch cost goes (130 ms -> a few ms per
iteration with eight runs side by side); the noise, the annealed prior log-odds, Adam's step size and the batch's record
indices are device tensors the host refreshes between replays.  --no-graph runs the same step eagerly on the same noise: the
same first loss to the last bit (tests/test_twin_tool.py on the GPU box); the gradients of either mode carry autograd's
scatter-add atomics, i.e. their own realisation of the out-of-range residue on every run.  --deadline-min stops early, still printing a last evaluation."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))
import numpy as np, torch
from oracle import air_oracle as ao, air_oracle_torch as at
from multi_mnist import generate_dataset, shift_zero_digits_images


def main(argv):
    argv = list(argv)
    use_graph = "--no-graph" not in argv
    if not use_graph:
        argv.remove("--no-graph")
    deadline = None
    if "--deadline-min" in argv:
        i = argv.index("--deadline-min")
        deadline = float(argv[i + 1]) * 60.0
        del argv[i:i + 2]
    eval_every = 5000
    if "--eval-every" in argv:
        i = argv.index("--eval-every")
        eval_every = int(argv[i + 1])
        del argv[i:i + 2]
    seed, iters = int(argv[0]), int(argv[1])
    torch.set_default_device("cuda")
    hp = dict(ao.TRAINING_HP)
    bg = None
    if len(argv) > 2:                                      # clutter background, already scaled (tests/golden/backgrounds.npz)
        f_, key_ = argv[2].rsplit(":", 1)
        bg = np.load(f_)[key_].astype(np.float32)
        if len(argv) > 3 and bg.max() > 0:                 # rescaled to a maximum intensity, as training.py --bg-max-intensity does
            bg = bg / bg.max() * min(float(argv[3]), 1.0)
    ds = generate_dataset(bg=bg)
    te_im, te_dg = shift_zero_digits_images(ds["test_images"], ds["test_digits"])
    tr_im, tr_dg = torch.tensor(ds["train_images"]), torch.tensor(ds["train_digits"].astype(np.int32))
    te_im, te_dg = torch.tensor(np.ascontiguousarray(te_im)), torch.tensor(np.ascontiguousarray(te_dg).astype(np.int32))
    params = {k: torch.tensor(v, requires_grad=True) for k, v in ao.init_params(hp, seed).items()}
    m = {k: torch.zeros_like(p) for k, p in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    g = torch.Generator(device="cuda").manual_seed(seed)   # noise and batch order: drawn OUTSIDE the graph, into static tensors
    N, Z, d, B = hp["max_steps"], hp["vae_latent_dimensions"], hp["windows_size"] ** 2, 64
    sched = ao.TRAINING_ANNEALING["z_pres_prior_log_odds"]
    c = hp["gradient_clipping_norm"]

    def noise(b):
        return dict(eps_scale=torch.randn(N, b, 1, generator=g), eps_shift=torch.randn(N, b, 2, generator=g),
                    eps_z=torch.randn(N, b, Z, generator=g), eps_x=torch.randn(N, b, d, generator=g), u=torch.rand(N, b, generator=g))

    def evaluate(lo):
        with torch.no_grad():
            o = at.air_forward(params, te_im, te_dg, noise(len(te_im)), hp, False, lo)
        dig = o["rec_num_digits"]
        acc = [float((dig[te_dg == k] == k).float().mean()) for k in range(3)]
        return float((dig == te_dg).float().mean()), acc

    # what changes from iteration to iteration, as device tensors
    idx = torch.zeros(B, dtype=torch.int64)
    nz = noise(B)
    lo_t = torch.zeros(())
    lr_t = torch.zeros(())
    loss_t = torch.zeros(())
    inv_c = torch.tensor(1.0 / c)

    def step():
        out, grads = at.loss_and_grads(params, tr_im[idx], tr_dg[idx], nz, hp, lo_t)
        loss_t.copy_(out["loss"].detach())
        gn = torch.sqrt(sum((x.detach() ** 2).sum() for x in grads.values()))
        scale = c * torch.minimum(1.0 / gn, inv_c)
        with torch.no_grad():
            for k, p in params.items():
                gk = grads[k] * scale
                m[k] += (gk - m[k]) * 0.1
                v[k] += (gk * gk - v[k]) * 0.001
                p -= (m[k] * lr_t) / (torch.sqrt(v[k]) + 1e-8)

    def set_iteration(it, batch):
        idx.copy_(batch)
        for k, fresh in noise(B).items():
            nz[k].copy_(fresh)
        lo_t.fill_(float(ao.annealed_value(sched, it)))
        t = it + 1
        lr_t.fill_(hp["learning_rate"] * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t))

    t0 = time.time()
    graph = None
    if use_graph:
        # the constants the op sequence caches (air_oracle_torch._const) and the lazily initialised libraries exist before the
        # capture: one throw-away forward + backward on a side stream (no update: lr = 0, and m / v are restored)
        snap = {k: (p.detach().clone(), m[k].clone(), v[k].clone()) for k, p in params.items()}
        set_iteration(0, torch.arange(B))
        lr_t.zero_()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, p in params.items():
                p.copy_(snap[k][0]); m[k].copy_(snap[k][1]); v[k].copy_(snap[k][2])
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
    g.manual_seed(seed)                                    # (the throw-away passes drew noise: both modes start from the same state)
    perm = torch.randperm(len(tr_im), generator=g)
    ptr = 0
    stopped, first_loss = None, None
    for it in range(iters):
        if it % eval_every == 0:
            a, acc = evaluate(float(ao.annealed_value(sched, it)))
            print(json.dumps({"seed": seed, "step": it, "accuracy": round(a, 3), "acc012": [round(x, 2) for x in acc],
                              "wall_s": round(time.time() - t0, 1)}), flush=True)
            if deadline is not None and time.time() - t0 > deadline:
                stopped = it
                break
        if ptr + B > len(tr_im):
            perm, ptr = torch.randperm(len(tr_im), generator=g), 0
        set_iteration(it, perm[ptr:ptr + B]); ptr += B
        if graph is not None:
            graph.replay()
        else:
            step()
        if it == 0:
            first_loss = float(loss_t)
    if stopped is None:
        a, acc = evaluate(float(ao.annealed_value(sched, iters)))
        print(json.dumps({"seed": seed, "step": iters, "accuracy": round(a, 3), "acc012": [round(x, 2) for x in acc],
                          "wall_s": round(time.time() - t0, 1), "final": True}), flush=True)
    return params, first_loss


if __name__ == "__main__":
    main(sys.argv[1:])
