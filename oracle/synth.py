"""Synthetic multi-object canvases for oracle/parity tests -- TEST INFRASTRUCTURE.

A cheap stand-in for the multi-MNIST generator (multi_mnist.py:82-183): 0..max
random-ink blobs of 14..20 px placed without pixel overlap on a C x C canvas
("~85 % zeros + Uniform(0,1) blobs").  Deterministic in `seed`."""
import numpy as np


def blob_canvases(batch, canvas=50, max_digits=2, seed=0, dtype=np.float32):
    rng = np.random.RandomState(seed)
    imgs = np.zeros((batch, canvas, canvas), dtype)
    counts = rng.randint(0, max_digits + 1, size=batch).astype(np.int32)
    for b in range(batch):
        placed = 0
        tries = 0
        while placed < counts[b] and tries < 200:
            tries += 1
            h, w = rng.randint(14, 21, size=2)
            y, x = rng.randint(0, canvas - h + 1), rng.randint(0, canvas - w + 1)
            if imgs[b, y:y + h, x:x + w].max() > 0:
                continue
            yy, xx = np.mgrid[0:h, 0:w]
            r = np.hypot((yy - h / 2 + 0.5) / (h / 2), (xx - w / 2 + 0.5) / (w / 2))
            ink = (np.abs(r - 0.6) < 0.22) * rng.uniform(0.5, 1.0, size=(h, w))
            imgs[b, y:y + h, x:x + w] = ink.astype(dtype)
            placed += 1
        counts[b] = placed
    return imgs.reshape(batch, canvas * canvas), counts
