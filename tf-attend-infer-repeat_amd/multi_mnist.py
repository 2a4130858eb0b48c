"""Multi-digit canvas generator -- restatement of the DEFAULT path of the reference's
multi_mnist.py (generate_multi_image :82-183 with use_pixel_overlap=True, the __main__
driver :299-414) in numpy, writing .npz instead of TFRecords.

Glyph source: real MNIST idx files are used if present under ``mnist_data/`` (the reference
downloads them, :336 -- impossible here, no network); otherwise the 1 797 8x8 glyphs of
sklearn's ``load_digits`` (committed as data/digits8x8.npz), up-sampled (cubic) to 16x16 inside a
28x28 frame with the ink ramp stretched so that strokes saturate like MNIST's (the rendering
that trains most reliably of those tried, profiles/r01_glyph_sweep_60k.jsonl).  Everything downstream (cropping, rejection sampling of
non-overlapping positions, strata of 0..max_digits digits, shuffling, 1 000-image test split)
follows the reference.

  python multi_mnist.py [--max-digits 2] [--images-per-digit 20000] [--test-set-size 1000]
                        [--digit-gap 0] [--canvas-margin 0] [--bg-path X --bg-max-intensity 1.0]
"""
import argparse
import gzip
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CANVAS_SIZE = 50
IMAGE_SIZE = 28
MNIST_FOLDER = "mnist_data/"
MULTI_MNIST_FOLDER = "multi_mnist_data/"


def load_glyphs():
    """[n, 28*28] float32 in [0,1] + labels; MNIST if available, else up-sampled 8x8 digits."""
    for name in ("train-images-idx3-ubyte", "train-images-idx3-ubyte.gz"):
        p = os.path.join(MNIST_FOLDER, name)
        if os.path.exists(p):
            op = gzip.open if p.endswith(".gz") else open
            with op(p, "rb") as f:
                _, n, r, c = struct.unpack(">IIII", f.read(16))
                imgs = np.frombuffer(f.read(), np.uint8).reshape(n, r * c).astype(np.float32) / 255.0
            lp = p.replace("images-idx3", "labels-idx1")
            with op(lp, "rb") as f:
                f.read(8)
                labels = np.frombuffer(f.read(), np.uint8).astype(np.int64)
            return imgs, labels, "mnist"
    import scipy.ndimage as nd
    d = np.load(os.path.join(HERE, "data", "digits8x8.npz"))
    small = d["images"].astype(np.float32) / 16.0
    out = np.zeros((small.shape[0], IMAGE_SIZE, IMAGE_SIZE), np.float32)
    zoom = float(os.environ.get("AIR_GLYPH_ZOOM", "2.0"))                 # 8x8 -> 16x16: MNIST-like stroke scale
    order = int(os.environ.get("AIR_GLYPH_ORDER", "3"))                   # spline order of the up-sampling
    lo, hi = (float(v) for v in os.environ.get("AIR_GLYPH_CONTRAST", "0.25,0.65").split(","))   # ink ramp: MNIST strokes saturate
    for i, g in enumerate(small):
        big = np.clip(nd.zoom(g, zoom, order=order), 0.0, 1.0)
        big = np.clip((big - lo) / (hi - lo), 0.0, 1.0)
        big = np.where(big >= 0.15, big, 0.0)
        o = (IMAGE_SIZE - big.shape[0]) // 2
        out[i, o:o + big.shape[0], o:o + big.shape[1]] = big
    return out.reshape(-1, IMAGE_SIZE * IMAGE_SIZE), d["labels"].astype(np.int64), "digits8x8"


def read_image(path, max_intensity):
    """multi_mnist.py:17-33 (backgrounds)."""
    from PIL import Image
    image = np.asarray(Image.open(path).convert("L"), dtype=np.float32) / 255.0
    img_min, img_max = image.min(), image.max()
    if img_min != img_max:
        if img_min > 0.0:
            image = image - img_min
        if img_max > 0.0:
            image = image / img_max
        if max_intensity < 1.0:
            image = image * max_intensity
    else:
        if img_max > max_intensity:
            image = np.ones_like(image) * max_intensity
    return image


def crop_non_empty(image):
    """:36-43"""
    cols = np.nonzero(np.sum(image, axis=0))[0]
    rows = np.nonzero(np.sum(image, axis=1))[0]
    return image[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1]


def pixels_overlap(canvas, image, x, y):
    """:61-65"""
    h, w = image.shape
    window = canvas[y:y + h, x:x + w]
    return not np.array_equal(np.maximum(image, window), image + window)


class Generator:
    """State of the reference's module-level globals (digit_ids, next_digit_id :82-85, :343-346)."""

    def __init__(self, glyphs, rng):
        self.glyphs = glyphs
        self.rng = rng
        self.digit_ids = rng.permutation(len(glyphs))
        self.next = 0

    def multi_image(self, num_images, canvas_dim=CANVAS_SIZE, image_dim=IMAGE_SIZE, bg=None, margin=0):
        """generate_multi_image :82-183, default arguments (no scale/rotation jitter, gap 0,
        pixel-overlap rejection, up to 100 position attempts, restart the image on failure)."""
        rng = self.rng
        while True:
            canvas = np.zeros([canvas_dim, canvas_dim], np.float32)
            ids, positions, boxes = [], [], []
            if num_images == 0:
                break
            ok = True
            for i in range(num_images):
                idx = self.digit_ids[self.next]
                self.next += 1
                if self.next >= len(self.digit_ids):
                    self.digit_ids = rng.permutation(self.digit_ids)
                    self.next = 0
                image = crop_non_empty(self.glyphs[idx].reshape(image_dim, image_dim))
                h, w = image.shape
                found = False
                for _ in range(100):
                    x = rng.randint(margin, canvas_dim - w - margin + 1)
                    y = rng.randint(margin, canvas_dim - h - margin + 1)
                    found = True if i == 0 else not pixels_overlap(canvas, image, x, y)
                    if found:
                        break
                if not found:
                    ok = False
                    break
                canvas[y:y + h, x:x + w] += image
                positions.extend([x, y])
                boxes.extend([w, h])
                ids.append(idx)
            if ok:
                break
        if bg is not None:
            canvas = np.clip(canvas + bg, 0.0, 1.0)
        return canvas, ids, positions, boxes


def generate_dataset(max_digits=2, images_per_digit=20000, test_set_size=1000, seed=0, bg=None, margin=0,
                     canvas_dim=CANVAS_SIZE, verbose=False):
    """__main__ :341-413: strata of 0..max_digits digits, shuffled together, first
    `test_set_size` images -> test, the rest -> train.  Returns dict of arrays."""
    glyphs, labels, source = load_glyphs()
    rng = np.random.RandomState(seed)                                   # np.random.seed(0) :341
    gen = Generator(glyphs, rng)
    images, digits = [], []
    for nd_ in range(max_digits + 1):
        for item in range(images_per_digit):
            img, *_ = gen.multi_image(nd_, canvas_dim=canvas_dim, bg=bg, margin=margin)
            images.append(img.reshape(-1))
            digits.append(nd_)
            if verbose and (item + 1) % 5000 == 0:
                print("%d digits: %d done" % (nd_, item + 1), flush=True)
    images = np.stack(images).astype(np.float32)
    digits = np.asarray(digits, np.int32)
    perm = rng.permutation(len(images))                                 # shuffle_lists :215-225
    images, digits = images[perm], digits[perm]
    return dict(train_images=images[test_set_size:], train_digits=digits[test_set_size:],
                test_images=images[:test_set_size], test_digits=digits[:test_set_size], source=source)


def shift_zero_digits_images(images, digits):
    """read_test_data(..., shift_zero_digits_images=True) :284-294: one empty image first,
    then all non-empty ones, then the remaining empty ones."""
    empty = [i for i in range(len(digits)) if digits[i] == 0]
    non_empty = [i for i in range(len(digits)) if digits[i] > 0]
    if not empty:
        return images, digits
    order = [empty[0]] + non_empty + empty[1:]
    return images[order], digits[order]


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--max-digits", type=int, choices=list(range(7)), default=2)
    parser.add_argument("--images-per-digit", type=int, default=20000)
    parser.add_argument("--test-set-size", type=int, default=1000)
    parser.add_argument("--canvas-margin", type=int, default=0)
    parser.add_argument("--canvas-size", type=int, default=CANVAS_SIZE)
    parser.add_argument("--bg-path", default="")
    parser.add_argument("--bg-max-intensity", type=float, default=1.0)
    args = parser.parse_args()
    os.makedirs(MULTI_MNIST_FOLDER, exist_ok=True)
    bg = read_image(args.bg_path, args.bg_max_intensity) if args.bg_path else None
    ds = generate_dataset(args.max_digits, args.images_per_digit, args.test_set_size, bg=bg,
                          margin=args.canvas_margin, canvas_dim=args.canvas_size, verbose=True)
    np.savez(MULTI_MNIST_FOLDER + "common.npz", images=ds["train_images"], digits=ds["train_digits"])
    np.savez(MULTI_MNIST_FOLDER + "test.npz", images=ds["test_images"], digits=ds["test_digits"])
    print("glyph source: %s; wrote %d train / %d test images" % (ds["source"], len(ds["train_images"]), len(ds["test_images"])))
