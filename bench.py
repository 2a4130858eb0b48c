#!/usr/bin/env python3
"""bench.py -- AIR train-step throughput on MI355X (the BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--precision bf16|fp32] [--no-graph]
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" = one full AIR train step on one batch of synthetic 50x50 multi-object
canvases resident in HBM: Philox noise + annealing, hoisted x.Wx, 3 x (LSTM,
heads, glimpse read, VAE, canvas write), ELBO, full backward, weight grads,
[all-reduce], global-norm clip, TF-style Adam.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

T_START = time.perf_counter()          # phase_wall_s.import: torch + the extension load on a fresh box

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tf-attend-infer-repeat_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# training.py:100-122 (the benchmark configuration)
HP = dict(max_steps=3, max_digits=2, rnn_units=256, canvas_size=50, windows_size=28,
          vae_latent_dimensions=50, vae_recognition_units=(512, 256), vae_generative_units=(256, 512),
          scale_prior_mean=-1.0, scale_prior_variance=0.05, shift_prior_mean=0.0, shift_prior_variance=1.0,
          vae_prior_mean=0.0, vae_prior_variance=1.0, vae_likelihood_std=0.3,
          scale_hidden_units=64, shift_hidden_units=64, z_pres_hidden_units=64,
          z_pres_prior_log_odds=-0.01, z_pres_temperature=1.0, stopping_threshold=0.99,
          learning_rate=1e-4, gradient_clipping_norm=1.0)
ANNEAL = {"z_pres_prior_log_odds": {"init": 10000.0, "min": 0.000000001, "factor": 0.1, "iters": 3000,
                                    "staircase": False, "log": True}}
MIN_REPLAYS = 5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TF = {"bf16": 2500.0, "fp32": 157.3}


def synthetic_canvases(batch, canvas, max_digits, seed):
    """0..max_digits ring-shaped ink blobs (14..20 px) without overlap -- a stand-in for
    multi-MNIST canvases of the same sparsity."""
    rng = np.random.RandomState(seed)
    imgs = np.zeros((batch, canvas, canvas), np.float32)
    counts = rng.randint(0, max_digits + 1, size=batch).astype(np.int32)
    for b in range(batch):
        placed, tries = 0, 0
        while placed < counts[b] and tries < 200:
            tries += 1
            h, w = rng.randint(14, 21, size=2)
            y, x = rng.randint(0, canvas - h + 1), rng.randint(0, canvas - w + 1)
            if imgs[b, y:y + h, x:x + w].max() > 0:
                continue
            yy, xx = np.mgrid[0:h, 0:w]
            r = np.hypot((yy - h / 2 + 0.5) / (h / 2), (xx - w / 2 + 0.5) / (w / 2))
            imgs[b, y:y + h, x:x + w] = ((np.abs(r - 0.6) < 0.22) * rng.uniform(0.5, 1.0, size=(h, w))).astype(np.float32)
            placed += 1
        counts[b] = placed
    return imgs.reshape(batch, canvas * canvas), counts


def per_kernel_times(model, iters):
    """HIP-event pair around every kernel launch of the train step, on the stream the
    kernels are launched on (eager pass, same buffers as the timed region)."""
    ops = [(o.name, o) for o in model.train_step_ops()]
    s = model._stream()
    # all iterations are enqueued before the single synchronize: the host runs ahead of the device,
    # so an interval does not contain the host-side cost of a launch (the weight-gradient call
    # fills a 1 KB descriptor table); the per-op MEDIAN over iterations drops the ramp-up ones
    all_evs = []
    for _ in range(iters):
        evs = []
        for i, (name, op) in enumerate(ops):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op(s)
            e1.record()
            evs.append((e0, e1))
        all_evs.append(evs)
    torch.cuda.synchronize()
    acc = []
    for i in range(len(ops)):
        ts = sorted(evs[i][0].elapsed_time(evs[i][1]) * 1e3 for evs in all_evs)           # us
        acc.append(ts[len(ts) // 2] * iters)
    # Each interval also contains the launch gap and the processing of its closing event record.
    # That per-launch overhead is calibrated live: the same eager step WITHOUT inner events, timed by
    # one outer event pair, is the sum of the kernels' in-situ durations (the stream never idles:
    # 27 launches take the host ~80 us, the device ~200 us); the excess of the inner intervals over
    # it, spread evenly over the launches, is subtracted.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        for _, op in ops:
            op(s)
    e0.record()
    for _ in range(iters):
        for _, op in ops:
            op(s)
    e1.record()
    torch.cuda.synchronize()
    step_us = e0.elapsed_time(e1) * 1e3 / iters
    overhead = max(0.0, (sum(acc) / iters - step_us) / len(ops))
    acc = [max(a - overhead * iters, 0.0) for a in acc]
    out = {}
    for (name, op), t in zip(ops, acc):
        # grouped by kernel FUNCTION, the way `rocprofv3 --kernel-trace --stats` groups them
        d = out.setdefault(op.kernel, dict(us=0.0, launches=0, nbytes=0, flops=0, ops=[]))
        d["us"] += t / iters
        d["launches"] += 1
        d["nbytes"] += getattr(op, "nbytes", 0)
        d["flops"] += getattr(op, "flops", 0)
        d["ops"].append(name)
    out["__meta__"] = dict(event_overhead_us=round(overhead, 2), eager_step_us=round(step_us, 1))
    return out


def source_sha16():
    """sha256 of the kernel sources (csrc/*, include/*.h): profiles are only valid for the build they measured"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "tf-attend-infer-repeat_amd", "csrc", "*")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# What actually holds each kernel function (DESIGN.md sections 5, 8, 10).  The schema's two bounds are "hbm" and "mfma";
# where neither is the limiter the line says what is, next to the HBM fraction it still reports.
def bound_of(kernel):
    if kernel.startswith("write_bwd_graph_kernel"):
        return "lds-atomic-pipe"      # one sequential fp32 accumulator per corner slot on ds_add_f32 (4 cycles per term)
    if kernel.startswith("write_bwd_carried_kernel"):
        return "latency"              # 16 register chains of <= C*C/16 dependent fp32 adds per corner and tap + the term pass
    if kernel.startswith(("adam", "grad_sqnorm")):
        return "hbm"
    if kernel.startswith(("gemm", "wgrad", "bottleneck", "lstm")):
        return "hbm"                  # M = 64 / 192 rows: operand streaming, MFMA busy < 1 %
    return "latency"                  # pointwise / sampler launches of a few us: dependent-launch floor + one pass over the items


_PROFILE = None


def kernel_profile(kernel):
    """rocprofv3 numbers for `kernel` from the committed profile round (profiles/kernel_profile.json,
    written by tools/profile_round.sh + profile_merge.py: --kernel-trace --stats average duration,
    FETCH_SIZE / WRITE_SIZE traffic from separate --pmc passes, SQ_VALU_MFMA_BUSY_CYCLES).  PMC
    counters cannot be collected from inside this process.  A profile taken from other kernel
    sources than the ones built now is refused (stale=True, no numbers)."""
    global _PROFILE
    if _PROFILE is None:
        path = os.path.join(ROOT, "profiles", "kernel_profile.json")
        _PROFILE = json.load(open(path)) if os.path.exists(path) else {}
        _PROFILE["_stale"] = bool(_PROFILE) and _PROFILE.get("source_sha16") != source_sha16()
    if not _PROFILE.get("kernels") or _PROFILE["_stale"]:
        return {"stale": True} if _PROFILE.get("_stale") else {}
    for k, v in _PROFILE["kernels"].items():
        if k.startswith(kernel):
            return v
    return {}


def cpu_baseline(batch, seconds=15.0):
    """The oracle's un-fused torch-CPU port of the reference op sequence, timed on this box's
    host cores on a bounded sample of the same workload (same config, synthetic canvases)."""
    from oracle import air_oracle as ao
    from oracle import air_oracle_torch as at
    # the port is a chain of small un-fused ops: beyond a socket's worth of threads it only gets
    # slower (256 threads on the GPU box: 135 s per step), so the thread count is capped and reported
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    hp = dict(ao.TRAINING_HP)
    images, targets = synthetic_canvases(batch, hp["canvas_size"], hp["max_digits"], 0)
    tr = at.CpuTrainer(ao.init_params(hp, 0), hp)
    im, tg = torch.tensor(images), torch.tensor(targets)
    for _ in range(2):
        tr.step(im, tg, 9.21)
    times, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < seconds and len(times) < 200:
        t1 = time.perf_counter()
        tr.step(im, tg, 9.21)
        times.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    n = len(times)
    med = sorted(times)[n // 2]
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # value: from the MEDIAN step time (SURVEY 8(d) protocol); `cores` = the threads actually used, which is the cap --
    # NOT the box's core count, which is `cores_available`
    return dict(value=round(batch / med, 1), unit="images/sec", cores=threads, kind="port",
                threads=threads, thread_cap=16, cores_available=os.cpu_count(), cpu_model=cpu_model,
                median_step_ms=round(med * 1e3, 2), mean_step_ms=round(dt / n * 1e3, 2), steps_timed=n,
                sample="%d train steps of batch %d (%.1f s) of oracle/air_oracle_torch.CpuTrainer on %d torch threads (capped: "
                       "beyond a socket's worth of threads this chain of small ops only slows down) of %s logical cores, %s; "
                       "median step" % (n, batch, dt, threads, os.cpu_count(), cpu_model))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL)
    and relay rank 0's JSON line.  The parent never touches the GPU (device_count() does not
    initialise it) and never execs; it exits non-zero if any rank fails."""
    import subprocess
    have = torch.cuda.device_count()
    if os.environ.get("AIR_BENCH_SAME_DEVICE") == "1":
        have = max(have, n) if have >= 1 else 0      # test mode: all ranks share cuda:0 (gloo backend)
    if have < n:
        sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to report a "
                         "%d-GPU number from fewer devices\n" % (n, have, n))
        return 3
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every rank: if one dies early (bad device, RCCL init failure) the others would sit in the rendezvous or a
    # collective until the store / NCCL timeout -- terminate them instead and report the failure at once
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("AIR_BENCH_RANK_TIMEOUT_S", "3000"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = "rank exit codes %r" % rcs
        elif time.time() > deadline:
            failed = "ranks still running after the deadline (exit codes so far %r)" % rcs
        if failed or all(rc == 0 for rc in rcs):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in out_chunks if c).decode())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: %s\n" % failed)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--precision", default=os.environ.get("AIR_GEMM_PRECISION", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE: 64)")
    ap.add_argument("--workload", default="configs[1]", choices=["configs[1]", "configs[3]"],
                    help="configs[1]: 50x50, N=3, b=64 (the metric); configs[3]: stress, 128x128, 0-4 objects, N=5, b=256")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-steps", type=int, default=100, help="train steps captured per hipGraph replay (1 GPU), at most "
                    "(20 / 50 / 100 / 200 per replay: 0.1785 / 0.1780 / 0.1778 / 0.1776 ms per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary blocks (fp32 step, stress config, inference)")
    ap.add_argument("--backward", default=None,
                    help="sampler-backward order: reference | reference_carried | exact, or a schedule FIRST>SECOND@N (AIRModel(backward=(FIRST, "
                         "SECOND, N))).  Default: what training.py runs (air_model.TRAINING_BACKWARD).  With a schedule the line's "
                         "headline is the step after the switch and `init_phase` holds the step before it")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)"
                 % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_dp.py): the multi-rank flow on ONE device over gloo -- RCCL refuses two
    # ranks on one GPU.  The numbers of such a run are not a scaling measurement and are labelled so.
    same_device = os.environ.get("AIR_BENCH_SAME_DEVICE") == "1"
    backend = os.environ.get("AIR_BENCH_BACKEND", "nccl")
    if same_device:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from air import air_model as am
    from air import _hip as H
    phase = {"import": round(time.perf_counter() - T_START, 3)}
    backward = am.TRAINING_BACKWARD if args.backward is None else args.backward
    if isinstance(backward, str) and ">" in backward:
        first, rest = backward.split(">", 1)
        second, at = rest.split("@", 1)
        backward = (first, second, int(at))
    backward_name = backward if isinstance(backward, str) else "%s>%s@%d" % tuple(backward)
    B = args.batch
    hp = dict(HP)
    if args.workload == "configs[3]":
        hp.update(canvas_size=128, max_steps=5, max_digits=4)
        B = 256 if args.batch == 64 else args.batch
    images, targets = synthetic_canvases(B, hp["canvas_size"], hp["max_digits"], seed=1000 + rank)
    images_d, targets_d = torch.tensor(images, device=dev), torch.tensor(targets, device=dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def all_ok(ok):
        """every rank takes the same branch: MIN of a success flag over the ranks (a rank that alone skipped or re-captured
        would issue another collective sequence than its peers -- an RCCL hang instead of an error line)"""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t))

    def build_and_time(scope, exchange, optional=False):
        """one model, W warm-up + exactly K timed train steps (barrier + synchronize on both sides, max over ranks).
        optional: a failure to build / capture on ANY rank makes every rank return None."""
        model, err = None, None
        try:
            model = am.AIRModel(images_d, targets_d, cnn=False, train=True, scope=scope, annealing_schedules=ANNEAL, seed=0,
                                noise_seed=rank, gemm_precision=args.precision, backward=backward, dp_exchange=exchange, **hp)
        except Exception as e:
            if not optional or world == 1:
                raise
            err = e
        if optional and not all_ok(err is None):
            if err is not None:
                sys.stderr.write("bench.py: rank %d: the %r exchange model could not be built (%r); skipped on all ranks\n" % (rank, exchange, err))
            return None
        return _time_model(model)

    def _time_model(model):
        if world > 1:
            model.sync_parameters()            # replicas start (and, with one shared all-reduce, stay) identical
        # several train steps per hipGraph replay: amortises the replay's own launch cost.  Data parallel over RCCL: the
        # collective is captured between backward and optimizer, so the protocol is the 1-GPU one
        gsteps, mode = 1, "eager"
        if not args.no_graph:
            # the largest divisor of the step count up to --graph-steps: exactly args.steps steps are timed
            want = max(g for g in range(1, max(1, args.graph_steps) + 1) if args.steps % g == 0)
            if world > 1 and not model._collectives_capturable():
                want = 1
            err = None
            try:
                model.capture_graph(steps=want)
                gsteps = want
                if world == 1:
                    mode = "one hipGraph replay = %d train steps" % gsteps
                elif model._graph[1] is None:
                    mode = "one hipGraph replay = %d x (fwd+bwd -> gradient exchange -> clip+Adam)" % gsteps
                else:
                    mode = "[fwd+bwd graph] -> collective -> [clip+Adam graph]"
            except Exception as e:                                   # never lose the line over the capture of a collective
                if world == 1:
                    raise
                err = e
            if not all_ok(err is None):                              # one rank failed: ALL ranks fall back together
                sys.stderr.write("bench.py: rank %d: capturing the collective failed on %s (%r); collective between two graphs\n"
                                 % (rank, "this rank" if err is not None else "another rank", err))
                os.environ["AIR_DP_GRAPH_COLLECTIVE"] = "0"
                torch.cuda.synchronize()
                model.release_graph()
                model.capture_graph(steps=1)
                gsteps, mode = 1, "[fwd+bwd graph] -> collective -> [clip+Adam graph] (in-graph capture failed)"
        def timed_region():
            """W warm-up steps, then exactly K timed steps between barrier + synchronize; max over ranks"""
            warm_replays, warm_eager = args.warmup // gsteps, args.warmup % gsteps
            if warm_replays == 0 and gsteps > 1 and args.warmup > 0:
                warm_replays, warm_eager = 1, 0          # the timed region must not hold the graph's first replay: one whole replay (>= W steps)
            for _ in range(warm_replays):
                model.training()
            for _ in range(warm_eager):                 # remainder of the warm-up: single eager steps
                model.training(eager=True)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps // gsteps):      # one replay = gsteps train steps: exactly args.steps steps
                model.training()
            sync()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t)
            return dt
        init_phase = None
        sched = model.backward_schedule
        if sched is not None:
            # A backward schedule (training.py's default): the step BEFORE the switch is timed first, from global_step 0 -- the
            # same W + K protocol --, then global_step is set to the switch iteration and the model changes its launch list (and
            # captures its graph again) by itself: what is timed below and reported as the headline is the step the driver
            # runs from iteration N to the end of training (276 250 iterations: >= 98 % of them for N = 5 000).
            if (args.warmup + args.steps + gsteps) >= sched[2]:
                raise SystemExit("bench.py: W + K steps reach the schedule's switch iteration %d" % sched[2])
            dt0 = timed_region()
            init_phase = {"backward": sched[0], "iterations": "global_step < %d" % sched[2], "ms_per_step": round(dt0 / args.steps * 1e3, 4),
                          "images_per_sec": round(world * B * args.steps / dt0, 1), "steps": args.steps, "warmup": args.warmup}
            model.store.istate[H.IST_GLOBAL_STEP] = sched[2]
            model._host_step = None
            model.training()                            # the switch: launch lists rebuilt, graph captured again (not timed)
            sync()
            assert model.backward == sched[1], model.backward
        dt = timed_region()
        # ... and the SAME K steps once more, outside the timed region, cut into >= MIN_REPLAYS replays with a HIP event
        # between them: the spread of the step time over the run (min / median / max per replay).  Not the headline: a
        # replay boundary costs ~18 us, which K / 5 steps amortise less well than K.
        spread = None
        if world == 1 and not args.no_graph:
            ok = [g for g in range(1, args.steps + 1) if args.steps % g == 0 and args.steps // g >= MIN_REPLAYS]
            if ok:
                g2 = max(ok)
                model.release_graph()
                model.capture_graph(steps=g2)
                model.training()
                n_rep = args.steps // g2
                marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_rep + 1)]
                torch.cuda.synchronize()
                marks[0].record()
                for i in range(n_rep):
                    model.training()
                    marks[i + 1].record()                   # (an event record between replays: no host wait)
                torch.cuda.synchronize()
                per = sorted(marks[i].elapsed_time(marks[i + 1]) / g2 for i in range(n_rep))
                spread = {"replays": n_rep, "steps_per_replay": g2, "min": round(per[0], 4), "median": round(per[n_rep // 2], 4),
                          "max": round(per[-1], 4), "unit": "ms per step, per replay (HIP events)",
                          "note": "a second pass of the same K steps after the timed region, in shorter replays; `value` is the "
                                  "timed region above (%d steps per replay)" % gsteps}
        return model, dt, gsteps, mode, spread, init_phase

    exchange = os.environ.get("AIR_DP_EXCHANGE") or "flat"
    phase["build_models"] = 0.0
    t_ph = time.perf_counter()
    model, dt, gsteps, graph_mode, spread, init_phase = build_and_time("air", exchange)
    phase["timed"] = round(dt, 3)
    phase["build_models"] = round(time.perf_counter() - t_ph - dt, 3)      # construction, capture and warm-up
    exchange_runs = {exchange: round(dt / args.steps * 1e3, 4)}
    if world > 1 and not os.environ.get("AIR_DP_EXCHANGE"):
        # both forms of the gradient exchange are complete train steps with the reference's semantics:
        # time the other one the same way (its own W + K steps) and report the line of the faster, naming it
        # (a rank that cannot build it makes ALL ranks skip it: build_and_time(optional=True))
        other = "factors"
        res = build_and_time("air_" + other, other, optional=True)
        if res is None:
            exchange_runs[other] = "skipped: a rank could not build it"
        else:
            m2, dt2, g2, mode2, spread2, init2 = res
            exchange_runs[other] = round(dt2 / args.steps * 1e3, 4)
            if dt2 < dt:
                model, dt, gsteps, graph_mode, exchange, spread, init_phase = m2, dt2, g2, mode2, other, spread2, init2
    loss = float(model.loss)
    replicas_identical = None
    if world > 1:
        # every rank applied the same all-reduced gradient to the same state: the parameters must be
        # bit-identical across ranks after the run (int64 sum of the raw words, min == max over ranks)
        chk = model.store.params.view(torch.int32).to(torch.int64).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(int(lo) == int(hi))

    # the step's only collective, timed on its own (all ranks take part): SURVEY 8(e) asks for its
    # duration and bus bandwidth against the 153 GB/s/link xGMI bound
    ar = None
    if world > 1:
        try:
            g = model.store.grads
            for _ in range(5):
                dist.all_reduce(g)
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dist.all_reduce(g)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            nbytes = g.numel() * 4
            ar = {"bytes": nbytes, "us": round(us, 1), "algbw_GBs": round(nbytes / us * 1e-3, 1),
                  "busbw_GBs": round(2.0 * (world - 1) / world * nbytes / us * 1e-3, 1),
                  "xgmi_link_peak_GBs": 153.0}
        except Exception as e:                      # never lose the bench line over the extra report
            ar = {"error": repr(e)}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        line = {
            "metric": "images/sec AIR train step, 50x50 multi-MNIST b=64 N=3" if args.workload == "configs[1]" else
                      "images/sec AIR train step, stress 128x128 b=256 N=5",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "accumulate_dtype": "fp32",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: AIR train step, 50x50 canvas, 0-2 objects, batch 64/GPU, 3 steps, "
                                    "256 LSTM, z=50 (training.py:100-122)") if args.workload == "configs[1]" else
                                   "configs[3]: stress, 128x128 canvas, 0-4 objects, batch %d/GPU, 5 steps" % B,
                       "global_batch": world * B,
                       "hipgraph": not args.no_graph, "steps_per_graph_replay": gsteps, "graph_mode": graph_mode,
                       "parallelism": "dp%d" % world,
                       "backward": backward_name,
                       "phase": ("steady: the step training.py runs from iteration %d on (>= 98 %% of its 276 250 iterations); the "
                                 "step before the switch is in init_phase" % model.backward_schedule[2])
                                if model.backward_schedule is not None else "single order"},
            "per_gpu_images_per_sec": round(value / world, 1), "final_loss": round(loss, 3),
            "ms_per_step_by_replay": spread,
        }
        if init_phase is not None:
            line["init_phase"] = init_phase
            line["init_phase_ms_per_step"] = init_phase["ms_per_step"]
        t_ph = time.perf_counter()
        if not args.no_roofline and world == 1:
            model.release_graph()
            kt = per_kernel_times(model, 20)
            meta = kt.pop("__meta__")
            total_us = sum(d["us"] for d in kt.values())
            def roof(name, d):
                avg_us = d["us"] / d["launches"]
                per_launch = d["nbytes"] / d["launches"]
                gbs = per_launch / avg_us * 1e-3 if avg_us > 0 else 0.0
                prof = kernel_profile(name)
                r = {"kernel": name, "bound": bound_of(name), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": prof.get("hbm_bytes_per_launch"),
                     "avg_us": round(avg_us, 2), "avg_us_rocprofv3": prof.get("avg_us"),
                     "launches_per_step": d["launches"],
                     "algorithmic_bytes": int(per_launch), "share_of_step": round(d["us"] / total_us, 3),
                     # achieved / frac / avg_us: HIP events in THIS run; traffic / avg_us_rocprofv3 / mfma.busy_*: rocprofv3
                     # passes of the same command on the same sources, committed (PMC counters cannot be read in-process)
                     "source": {"achieved": "live HIP events", "avg_us": "live HIP events",
                                "traffic": "committed profile", "avg_us_rocprofv3": "committed profile",
                                "profile": "profiles/kernel_profile.json", "profile_source_sha16": (_PROFILE or {}).get("source_sha16")}}
                if d["flops"]:
                    r["mfma"] = {"flops_per_launch": int(d["flops"] / d["launches"]),
                                 "frac_of_peak_from_flops": round(d["flops"] / d["launches"] / (avg_us * 1e-6) * 1e-12
                                                                  / MFMA_PEAK_TF[args.precision], 5) if avg_us > 0 else None,
                                 "busy_counter_util": prof.get("mfma_util"),        # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE*256*4)
                                 "busy_cycles_per_launch": prof.get("mfma_busy_cycles"),
                                 "source": {"busy_counter_util": "committed profile", "busy_cycles_per_launch": "committed profile"}}
                if prof.get("stale"):
                    r["profile_stale"] = True        # profiles/kernel_profile.json was taken from other sources
                if name.startswith("write_bwd_graph_kernel"):
                    r["note"] = ("the reference's UnsortedSegmentSum order needs ONE sequential fp32 accumulator per corner slot, "
                                 "ds_add_f32 delivers 4.0 cycles per term and the slowest workgroup owns up to 4*C*C terms (16.9 of its "
                                 "24 us at 50x50) -- DESIGN.md section 8; the step's HBM-bound kernel is the Adam launch (roofline_all)")
                elif name.startswith("write_bwd_carried_kernel"):
                    r["note"] = ("the same term streams in at most 16 chunks per slot and tap, one register chain per chunk "
                                 "(DESIGN.md section 10): as long as its term pass + the longest chunk of its heaviest item")
                return r
            # dominant kernel = the kernel function with the largest share of the step (what the
            # rocprofv3 stats table lists first); the other functions follow in `roofline_all`
            ranked = sorted(kt.items(), key=lambda kv: -kv[1]["us"])
            line["roofline"] = roof(*ranked[0])
            line["roofline_all"] = [roof(k, v) for k, v in ranked[1:8]]
            step_bytes = 40 * model.store.num_trainable + 4 * B * model.store.dims["D"]
            line["step_roofline"] = {"algorithmic_bytes": step_bytes,
                                     "achieved_GBs": round(step_bytes / (ms * 1e-3) * 1e-9, 1),
                                     "frac_of_hbm_peak": round(step_bytes / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4),
                                     "sum_kernel_us": round(total_us, 1), "eager_step_us": meta["eager_step_us"],
                                     "event_overhead_us_per_launch": meta["event_overhead_us"], "launches": sum(x["launches"] for x in kt.values()),
                                     "mfma_flops": sum(x["flops"] for x in kt.values()),
                                     "mfma_frac_of_peak": round(sum(x["flops"] for x in kt.values()) / (ms * 1e-3) * 1e-12
                                                                / MFMA_PEAK_TF[args.precision], 5)}
            line["kernels"] = {k: {"us_per_step": round(v["us"], 2), "launches": v["launches"], "ops": v["ops"]}
                               for k, v in ranked}
        if not args.no_extras and world == 1:
            # secondary metric of SURVEY 8(d): inference (train=False forward, z_pres rounded) on the same batch
            inf = am.AIRModel(model.input_images, model.target_num_digits, cnn=False, train=False, reuse=True,
                              scope="air", annealing_schedules=ANNEAL, seed=rank, gemm_precision=args.precision, **hp)

            def time_forward(n=200):
                for _ in range(20):
                    inf.forward()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    inf.forward()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / n
            us_eager = time_forward()
            inf.capture_graph()
            us_graph = time_forward()
            # the forward is a chain of 13 dependent launches either way: report the faster way of issuing it, name it
            us_best = min(us_graph, us_eager)
            line["inference"] = {"images_per_sec": round(B / us_best * 1e6, 1), "us_per_forward": round(us_best, 1),
                                 "us_per_forward_graph": round(us_graph, 1), "us_per_forward_eager": round(us_eager, 1),
                                 "launches": len(inf._fwd) + 1,
                                 "mode": ("one hipGraph replay per forward" if us_graph <= us_eager else "eager launches") + ", train=False"}
        if not args.no_extras and world == 1 and args.workload == "configs[1]":
            # the same step at the reference's own precision (fp32 operands, exact-fp32 MFMA) ...
            def secondary(tag, prec, hp2, B2, steps, backward=model.backward):
                im2, tg2 = synthetic_canvases(B2, hp2["canvas_size"], hp2["max_digits"], seed=2000)
                m2 = am.AIRModel(torch.tensor(im2, device=dev), torch.tensor(tg2, device=dev), cnn=False, train=True,
                                 scope=tag, annealing_schedules=ANNEAL, seed=0, gemm_precision=prec,
                                 backward=backward, **hp2)
                g2 = 1 if args.no_graph else 4
                if not args.no_graph:
                    m2.capture_graph(steps=g2)
                for _ in range(max(1, 8 // g2)):
                    m2.training()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps // g2):
                    m2.training()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                out = {"ms_per_step": round(dt2 / (steps // g2 * g2) * 1e3, 4),
                       "images_per_sec": round(B2 * (steps // g2 * g2) / dt2, 1), "batch": B2, "dtype": prec,
                       "steps": steps // g2 * g2, "final_loss": round(float(m2.loss), 3)}
                m2.release_graph()
                del m2
                return out
            line["fp32"] = secondary("air_fp32", "fp32", hp, B, 100)
            # ... and BASELINE configs[3], the stress configuration (128x128 canvas, 0-4 objects, N=5, b=256)
            hp3 = dict(HP, canvas_size=128, max_steps=5, max_digits=4)
            line["stress_configs3"] = secondary("air_stress", args.precision, hp3, 256, 40)
            line["stress_configs3"]["workload"] = "configs[3]: 128x128 canvas, 0-4 objects, batch 256, 5 steps"
            P3 = sum(int(np.prod(v)) for v in [(128 * 128 + 256, 1024)]) + (model.store.num_trainable - (2756 * 1024))
            line["stress_configs3"]["frac_of_hbm_peak"] = round(
                (40 * P3 + 4 * 256 * 128 * 128) / (line["stress_configs3"]["ms_per_step"] * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4)
            # the same two configurations under each sampler-backward order on its own (both bit for bit against their oracle
            # orders; the carried one is opt-in: faster, and over 48 seeds per precision it does not keep the reference order's
            # success rate, from the start or from any switch iteration tried -- DESIGN.md sections 10 and 11)
            line["backward_orders"] = {}
            for mode in ("reference", "reference_carried"):
                line["backward_orders"][mode] = {
                    "configs[1]": secondary("air_" + mode, args.precision, hp, B, 100, backward=mode)["ms_per_step"],
                    "configs[3]": (line["stress_configs3"]["ms_per_step"] if mode == model.backward else
                                   secondary("air_stress_" + mode, args.precision, hp3, 256, 40, backward=mode)["ms_per_step"])}
            line["backward_orders"]["unit"] = "ms per step (secondary blocks: 4 steps per replay, 100 / 40 steps)"
        phase["extras"] = round(time.perf_counter() - t_ph, 3)      # per-kernel events, inference, fp32 and stress blocks
        if not args.no_cpu_baseline and world == 1 and args.workload == "configs[1]":
            t_ph = time.perf_counter()
            line["cpu_baseline"] = cpu_baseline(B)
            phase["cpu_baseline"] = round(time.perf_counter() - t_ph, 3)
        line["phase_wall_s"] = phase
        if world > 1 and ar is not None:
            line["allreduce"] = ar
        if world > 1:
            # self-describing multi-GPU line: what actually carried the collective
            try:
                nccl_v = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
            except Exception:
                nccl_v = None
            line["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                   "rccl_version": nccl_v, "torch": torch.__version__, "hip": getattr(torch.version, "hip", None),
                                   "devices_visible": torch.cuda.device_count(),
                                   "collective": ("one all_reduce(SUM) of the flat fp32 gradient (+ loss/accuracy tail) per step"
                                                  if exchange == "flat" else
                                                  "all_gather of the dWx factors (X, sum_t dgates) + all_reduce(SUM) of the other "
                                                  "gradients (+ loss/accuracy tail) per step"),
                                   "gradient_exchange": exchange, "ms_per_step_by_exchange": exchange_runs}
            line["replicas_bit_identical"] = replicas_identical
            if same_device or backend != "nccl":
                line["test_mode"] = "ranks share one device over %s: NOT a scaling measurement" % backend
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
