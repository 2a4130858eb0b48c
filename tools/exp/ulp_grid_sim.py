"""CPU-only feasibility check: the reference's sequential fp32 sum as an INTEGER sum on the
accumulator's ulp grid.

While the accumulator stays inside one binade [2^e, 2^(e+1)) its ulp u = 2^(e-23) is constant and
    fl(acc + t) = acc + rne(t / u) * u          (no tie: t / u not exactly k + 1/2)
i.e. acc / u performs an integer addition -- associative, so a chunk of terms can be reduced in parallel to ONE integer
S = sum rne(t_i / u) (+ the min / max of its prefix sums for the range check) and the serial walk over a stream is one
integer add + one range check per chunk.  A chunk "validates" if (a) the binade guessed for it from a parallel estimate of
the accumulator at its start (the running fp32 sum of the chunk sums, as order="carried16" computes it) is the true one,
(b) every prefix inside the chunk stays in that binade with the same sign, (c) no term is a tie.  Other chunks fall back
to the sequential chain.  This script takes the corner streams of the write backward as the model makes them at
initialisation (the 2116-stream set of tests/test_graph_exec.py::test_carried_order_keeps_the_residue_of_real_corner_streams,
two batches of 64 blob canvases through the numpy forward), evaluates them that way for 16 / 32 / 64-term chunks, checks
bit equality with the sequential sum and reports which fraction of the terms sits in validating chunks.
  python tools/exp/ulp_grid_sim.py [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import air_oracle as ao          # noqa: E402
from oracle.synth import blob_canvases       # noqa: E402

f = np.float32


def corner_streams(seeds=2):
    hp = dict(ao.TRAINING_HP)
    for seed in range(seeds):
        B = 64
        images, targets = blob_canvases(B, 50, 2, seed=100 + seed)
        o = ao.air_forward(ao.init_params(hp, seed), images, targets, ao.make_noise(hp, B, seed + 50), hp, True, 9.21)
        r = o["_running_recon"]
        rc = np.clip(r, 0, 1).astype(f)
        p1, p0 = rc + f(ao.EPS), (f(1) - rc) + f(ao.EPS)
        g = np.where((r <= 1) & (np.minimum(r, 1) >= 0), -(f(1) / f(B)) * (images / p1 - (f(1) - images) / p0), 0).astype(f)
        for t in range(o["rec_scales"].shape[1]):
            s, xs, ys = o["rec_scales"][:, t, 0], o["rec_shifts"][:, t, 0], o["rec_shifts"][:, t, 1]
            th = np.zeros((B, 2, 3), f)
            th[:, 0, 0] = th[:, 1, 1] = f(1) / s
            th[:, 0, 2], th[:, 1, 2] = (-xs) / s, (-ys) / s
            _, aux = ao.transformer(o["rec_windows"][:, t].reshape(B, 28, 28).astype(f), th, (50, 50), return_aux=True)
            X, Y, x0, x1, y0, y1 = (aux[q] for q in ("x", "y", "x0", "x1", "y0", "y1"))
            gg = (o["_z_pres"][:, t][:, None] * g).astype(f)
            wx0, wx1, wy0, wy1 = x1.astype(f) - X, X - x0.astype(f), y1.astype(f) - Y, Y - y0.astype(f)
            for b in range(B):
                idx = [y0[b] * 28 + x0[b], y1[b] * 28 + x0[b], y0[b] * 28 + x1[b], y1[b] * 28 + x1[b]]
                val = [(wx0[b] * wy0[b]) * gg[b], (wx0[b] * wy1[b]) * gg[b], (wx1[b] * wy0[b]) * gg[b], (wx1[b] * wy1[b]) * gg[b]]
                for sl in (0, 27, 756, 783):
                    st = [v[i == sl] for i, v in zip(idx, val)]
                    if max(len(q) for q in st) <= 64:
                        continue
                    yield np.concatenate(st).astype(f)


def binade(x):
    """exponent e with 2^e <= |x| < 2^(e+1) (fp32 normal numbers); None for 0 / subnormal"""
    x = abs(float(x))
    if x < 2.0 ** -126:
        return None
    return int(np.floor(np.log2(x))) if not np.isinf(x) else None


def frexp_e(x):
    m, e = np.frexp(np.float32(abs(x)))
    return int(e) - 1 if m != 0 else None


def evaluate(stream, cs):
    """-> (result, terms in validating chunks, chunks validating, chunks) for chunk size cs"""
    n = len(stream)
    nch = -(-n // cs)
    # parallel estimate of the accumulator at every chunk start: running fp32 sum of the chunks' own sums
    P = np.zeros(nch + 1, f)
    for k in range(nch):
        ck = np.add.accumulate(np.concatenate([[f(0)], stream[k * cs:(k + 1) * cs]]).astype(f))[-1]
        P[k + 1] = f(P[k] + ck)
    acc = f(0)
    ok_terms = ok_chunks = 0
    for k in range(nch):
        ch = stream[k * cs:(k + 1) * cs]
        e_guess = frexp_e(P[k])
        e_true = frexp_e(acc)
        valid = e_guess is not None and e_true == e_guess and np.sign(P[k]) == np.sign(acc)
        if valid:
            u = 2.0 ** (e_guess - 23)
            q = ch.astype(np.float64) / u                       # exact (power of two)
            r = np.rint(q)
            if np.any(np.abs(q - np.trunc(q)) == 0.5) or np.any(np.abs(r) >= 2.0 ** 31):
                valid = False                                   # a tie (or an absurd term): the order of the parity matters
            else:
                A = float(acc) / u                              # integer in +-[2^23, 2^24)
                pre = A + np.cumsum(r)
                lo, hi = (2.0 ** 23, 2.0 ** 24) if A > 0 else (-(2.0 ** 24), -(2.0 ** 23))
                if A > 0:
                    valid = pre.min() >= lo and pre.max() < hi
                else:
                    valid = pre.min() > lo and pre.max() <= hi
                if valid:
                    acc = f(pre[-1] * u)
        if valid:
            ok_terms += len(ch)
            ok_chunks += 1
        else:
            acc = np.add.accumulate(np.concatenate([[acc], ch]).astype(f))[-1]
    return acc, ok_terms, ok_chunks, nch


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    streams = list(corner_streams(seeds))
    n_terms = sum(len(s) for s in streams)
    print("%d corner streams, %d terms (mean %.0f, max %d)" % (len(streams), n_terms, n_terms / len(streams), max(len(s) for s in streams)))
    seq = [np.add.accumulate(np.concatenate([[f(0)], s]).astype(f))[-1] for s in streams]
    for cs in (16, 32, 64):
        same = okt = okc = nc = 0
        worst = []
        for s, ref in zip(streams, seq):
            got, a, b, c = evaluate(s, cs)
            same += int(got == ref)
            okt += a; okc += b; nc += c
            worst.append(1.0 - a / len(s))
        worst = np.array(worst)
        # serial cost model of the walk: 1 unit per validating chunk, cs units (dependent fp32 adds) per fallback chunk
        print("chunks of %2d terms: bit-equal %d / %d; terms in validating chunks %.1f %% (chunks %.1f %%); per stream the share of "
              "fallback terms: median %.1f %%, p90 %.1f %%, max %.1f %%" % (cs, same, len(streams), 100.0 * okt / n_terms, 100.0 * okc / nc,
                                                                         100 * np.median(worst), 100 * np.percentile(worst, 90), 100 * worst.max()))


if __name__ == "__main__":
    main()
