"""Multi-digit canvas generator -- restatement of the DEFAULT path of the reference's
multi_mnist.py (generate_multi_image :82-183 with use_pixel_overlap=True, the __main__
driver :299-414) in numpy, writing .npz instead of TFRecords.

Glyph source: real MNIST idx files are used if present under ``mnist_data/`` (the reference
downloads them, :336 -- impossible here, no network); otherwise the 1 797 8x8 glyphs of
sklearn's ``load_digits`` (committed as data/digits8x8.npz), up-sampled (cubic) to 12x12 inside a
28x28 frame with the ink ramp stretched so that strokes saturate like MNIST's (the rendering
that trains most reliably of those tried: with 16x16 or larger glyphs most seeds stall on an
over-counting plateau, with 12x12 every fp32 seed ends at 96-99 % digit-count accuracy --
profiles/r01_glyph_sweep*_60k.jsonl, profiles/r01_seed_sweep_12px_120k.jsonl).  Everything downstream (cropping, rejection sampling of
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
    zoom = float(os.environ.get("AIR_GLYPH_ZOOM", "1.5"))                 # 8x8 -> 12x12: fits the scale prior (sigmoid(-1) * 50 = 13.4 px)
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


def add_buffer(image, buffer_width):
    """:44-58 -- every empty pixel within `buffer_width` (Chebyshev) of an ink pixel becomes 1:
    a dilation by a (2b+1)^2 square, used to keep a gap between digits under pixel overlap."""
    import scipy.ndimage as nd
    b = buffer_width
    near = nd.maximum_filter((image > 0).astype(np.uint8), size=2 * b + 1, mode="constant", cval=0) > 0
    return np.where((image == 0) & near, np.float32(1.0), image).astype(image.dtype)


def pixels_overlap(canvas, image, x, y):
    """:61-65"""
    h, w = image.shape
    window = canvas[y:y + h, x:x + w]
    return not np.array_equal(np.maximum(image, window), image + window)


def bounding_boxes_overlap(x, y, w, h, positions, boxes, gap):
    """:68-79, with the reference's exact tests: a candidate is rejected as soon as its
    gap-inflated x-extent intersects a placed box's x-extent (whatever the rows), and also in the
    (degenerate) case l1y >= r2y and l2y >= r1y."""
    l1x, l1y, r1x, r1y = x - gap, y - gap, x + w + gap - 1, y + h + gap - 1
    for i in range(len(positions) // 2):
        px, py = positions[2 * i], positions[2 * i + 1]
        bw, bh = boxes[2 * i], boxes[2 * i + 1]
        l2x, l2y, r2x, r2y = px, py, px + bw - 1, py + bh - 1
        if l1x <= r2x and l2x <= r1x:
            return True
        if l1y >= r2y and l2y >= r1y:
            return True
    return False


class Generator:
    """State of the reference's module-level globals (digit_ids, next_digit_id :82-85, :343-346)."""

    def __init__(self, glyphs, rng):
        self.glyphs = glyphs
        self.rng = rng
        self.digit_ids = rng.permutation(len(glyphs))
        self.next = 0

    def multi_image(self, num_images, canvas_dim=CANVAS_SIZE, image_dim=IMAGE_SIZE, bg=None,
                    min_w=1.0, max_w=1.0, min_h=1.0, max_h=1.0, min_ang=0.0, max_ang=0.0,
                    gap=0, margin=0, use_pixel_overlap=True):
        """generate_multi_image :82-183: scale / rotation jitter with order-5 splines (:117-139),
        rejection sampling of positions (up to 100 attempts, then the whole image is restarted)
        under pixel overlap (with an optional gap buffer) or bounding-box overlap."""
        import scipy.ndimage as nd
        rng = self.rng
        while True:
            canvas = np.zeros([canvas_dim, canvas_dim], np.float32)
            canvas_with_buffer = canvas
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
                if min_w != 1.0 or max_w != 1.0 or min_h != 1.0 or max_h != 1.0:
                    new_width = rng.uniform(min_w, max_w)
                    new_height = rng.uniform(min_h, max_h)
                    image = nd.affine_transform(image, matrix=np.array([[1.0 / new_height, 0.0], [0.0, 1.0 / new_width]]),
                                                output_shape=(int(image_dim * new_height), int(image_dim * new_width)),
                                                order=5)
                    image = np.clip(image, 0.0, 1.0)
                    image = crop_non_empty(np.where(image >= 0.05, image, np.zeros_like(image)))
                if min_ang != 0.0 or max_ang != 0.0:
                    image = nd.rotate(image, rng.uniform(min_ang, max_ang), order=5)
                    image = np.clip(image, 0.0, 1.0)
                    image = crop_non_empty(np.where(image >= 0.05, image, np.zeros_like(image)))
                h, w = image.shape
                if w + 2 * margin > canvas_dim or h + 2 * margin > canvas_dim:
                    ok = False                                   # (the reference's IndexError path :166)
                    break
                found = False
                for _ in range(100):
                    x = rng.randint(margin, canvas_dim - w - margin + 1)
                    y = rng.randint(margin, canvas_dim - h - margin + 1)
                    if i == 0:
                        found = True
                    elif use_pixel_overlap:
                        found = not pixels_overlap(canvas_with_buffer, image, x, y)
                    else:
                        found = not bounding_boxes_overlap(x, y, w, h, positions, boxes, gap)
                    if found:
                        break
                if not found:
                    ok = False
                    break
                canvas[y:y + h, x:x + w] += image
                if use_pixel_overlap and num_images > 1:
                    canvas_with_buffer = add_buffer(canvas, gap) if gap > 0 else canvas
                positions.extend([x, y])
                boxes.extend([w, h])
                ids.append(idx)
            if ok:
                break
        if bg is not None:
            canvas = np.clip(canvas + bg, 0.0, 1.0)
        return canvas, ids, positions, boxes


def generate_strata(max_digits=2, images_per_digit=20000, seed=0, bg=None, canvas_dim=CANVAS_SIZE, verbose=False,
                    **jitter):
    """__main__ :341-381: one stratum of `images_per_digit` canvases per digit count, with the
    metadata the reference stores (glyph indices, positions, boxes, labels)."""
    glyphs, labels, source = load_glyphs()
    rng = np.random.RandomState(seed)                                   # np.random.seed(0) :341
    gen = Generator(glyphs, rng)
    strata = []
    for nd_ in range(max_digits + 1):
        st = dict(images=[], indices=[], positions=[], boxes=[], labels=[], digits=[])
        for item in range(images_per_digit):
            img, ids, pos, box = gen.multi_image(nd_, canvas_dim=canvas_dim, bg=bg, **jitter)
            st["images"].append(img.reshape(-1))
            st["indices"].append(list(ids)); st["positions"].append(list(pos)); st["boxes"].append(list(box))
            st["labels"].append([int(labels[i]) for i in ids]); st["digits"].append(nd_)
            if verbose and (item + 1) % 5000 == 0:
                print("%d digits: %d done" % (nd_, item + 1), flush=True)
        strata.append(st)
    return strata, rng, source


def generate_dataset(max_digits=2, images_per_digit=20000, test_set_size=1000, seed=0, bg=None, margin=0,
                     canvas_dim=CANVAS_SIZE, verbose=False, **jitter):
    """__main__ :341-413: strata of 0..max_digits digits, shuffled together, first
    `test_set_size` images -> test, the rest -> train.  Returns dict of arrays."""
    strata, rng, source = generate_strata(max_digits, images_per_digit, seed, bg, canvas_dim, verbose,
                                          margin=margin, **jitter)
    images = np.stack([im for st in strata for im in st["images"]]).astype(np.float32)
    digits = np.asarray([d for st in strata for d in st["digits"]], np.int32)
    perm = rng.permutation(len(images))                                 # shuffle_lists :215-225
    images, digits = images[perm], digits[perm]
    return dict(train_images=images[test_set_size:], train_digits=digits[test_set_size:],
                test_images=images[:test_set_size], test_digits=digits[:test_set_size], source=source)


# ----------------------------------------------------------------------------- reference file format
def write_to_records(filename, images, indices, positions, boxes, labels, digits, side=None):
    """:186-212 -- `filename`.tfrecords with the reference's feature set (height, width, digits as
    int64; indices / positions / boxes / labels as raw int32 bytes; image as raw float32 bytes)."""
    import tfrecord
    recs = []
    for k in range(len(images)):
        img = np.asarray(images[k], np.float32)
        rows = cols = side if side is not None else (img.shape[0] if img.ndim == 2 else int(round(np.sqrt(img.size))))
        recs.append(tfrecord.encode_example({
            "height": ("int64", [rows]), "width": ("int64", [cols]), "digits": ("int64", [int(digits[k])]),
            "indices": ("bytes", [np.asarray(indices[k], np.int32).tobytes()]),
            "positions": ("bytes", [np.asarray(positions[k], np.int32).tobytes()]),
            "boxes": ("bytes", [np.asarray(boxes[k], np.int32).tobytes()]),
            "labels": ("bytes", [np.asarray(labels[k], np.int32).tobytes()]),
            "image": ("bytes", [np.ravel(img).tobytes()]),
        }))
    return tfrecord.write_records(filename + ".tfrecords", recs)


class ShuffleBatchQueue:
    """read_and_decode (:228-249) on the device: tf.train.shuffle_batch([image, digits], batch_size,
    capacity=10000 + 10 * batch_size, min_after_dequeue=10000) over the epoch-repeating record stream of ONE reader
    (training.py:76-81), the decoded records resident in HBM.  `images` [n, D] float32 / `digits` [n] int32 are device
    tensors; every batch lands in the caller's `out_images` [B, D] / `out_digits` [B] (the train model's input buffers).

    next_batch(): dequeue + gather on the current stream (two launches in front of the step that consumes the batch).
    graph_hooks(steps): the form for AIRModel.capture_graph(steps, between_steps=): the captured replay starts with ONE
    launch that makes the picks of all its `steps` batches into a device table (air_shuffle_batch_dequeue_many, the queue
    staged in LDS once) and holds one more launch per step, the row gather of that step's batch.  The batches are the same
    sequence either way, pick for pick the numpy model of the reference's queue that tests/test_shuffle_queue.py holds.
    (Measured on one MI355X, 50 steps per replay, ms per train step: no input work 0.1774; a gather per step 0.1801; this
    form 0.1815; a serial dequeue + gather in front of every step 0.186-0.190; a dequeue_many concurrent with the steps --
    on a forked branch of the graph or on a second stream -- 0.2085: DESIGN.md section 11.3.)"""

    def __init__(self, images, digits, batch_size, out_images, out_digits, seed=0, min_after_dequeue=10000):
        import ctypes as C
        import torch
        from air import _hip as H
        self._C, self._torch, self._H = C, torch, H
        dev = images.device
        self.images, self.digits, self.out_images, self.out_digits = images, digits, out_images, out_digits
        self.batch, self.D = int(batch_size), int(images.shape[1])
        self.capacity = min_after_dequeue + 10 * self.batch                       # :246
        self.queue = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.state = torch.zeros(2, dtype=torch.int64, device=dev)
        self.picks = torch.zeros(self.batch, dtype=torch.int32, device=dev)
        self._sq = H.ShuffleBatch(self.queue.data_ptr(), self.state.data_ptr(), self.picks.data_ptr(), self.capacity,
                                  self.batch, min_after_dequeue, int(images.shape[0]), seed)
        self._table = None          # [steps, batch] picks of the replay about to run (read by the captured gathers)
        H.check(H.lib().air_shuffle_batch_init(C.byref(self._sq), self._s()), "air_shuffle_batch_init")

    def _s(self):
        return self._C.c_void_p(self._torch.cuda.current_stream(self.images.device).cuda_stream)

    def _gather(self, picks):
        H = self._H
        H.check(H.lib().air_batch_gather(self.images.data_ptr(), self.digits.data_ptr(), picks.data_ptr(),
                                         self.out_images.data_ptr(), self.out_digits.data_ptr(), self.batch, self.D, self._s()),
                "air_batch_gather")

    def next_batch(self, _i=0):
        H = self._H
        if self._table is not None:
            raise RuntimeError("this queue feeds a captured graph (graph_hooks): batches come out of its replays")
        H.check(H.lib().air_shuffle_batch_dequeue(self._C.byref(self._sq), self._s()), "air_shuffle_batch_dequeue")
        self._gather(self.picks)

    def graph_hooks(self, steps):
        """-> (between_steps, after_steps) for AIRModel.capture_graph"""
        if steps < 1:
            raise ValueError("steps per replay must be positive")
        if self._table is None:
            self._table = self._torch.zeros(steps, self.batch, dtype=self._torch.int32, device=self.images.device)
        elif self._table.shape[0] != steps:
            raise ValueError("graph_hooks was set up for %d steps per replay" % self._table.shape[0])

        def between_steps(i):
            H = self._H
            if i == 0:                                   # the picks of the replay's batches: the replay's first launch
                H.check(H.lib().air_shuffle_batch_dequeue_many(self._C.byref(self._sq), steps, self._table.data_ptr(), self._s()),
                        "air_shuffle_batch_dequeue_many")
            self._gather(self._table[i])
        return between_steps, None


def read_test_data(filename, shift_zero_digits_images=False):
    """:254-296 -- reads a .tfrecords file written by the reference (or by write_to_records)."""
    import tfrecord
    images_list, digits_list = [], []
    indices_list, positions_list, boxes_list, labels_list = [], [], [], []
    for rec in tfrecord.read_records(filename):
        ex = tfrecord.parse_example(rec)
        n = int(ex["digits"][0])
        images_list.append(np.frombuffer(ex["image"][0], np.float32))
        digits_list.append(n)
        i32 = lambda k: np.frombuffer(ex[k][0], np.int32) if k in ex and ex[k] else np.zeros(0, np.int32)
        indices_list.append(i32("indices")[:n])
        positions_list.append(i32("positions")[:n * 2])
        boxes_list.append(i32("boxes")[:n * 2])
        labels_list.append(i32("labels")[:n])
    images_list, digits_list = np.array(images_list), np.array(digits_list)
    if shift_zero_digits_images:
        images_list, digits_list = _shift_zero(images_list, digits_list)
    return images_list, digits_list, indices_list, positions_list, boxes_list, labels_list


def shift_zero_digits_images(images, digits):
    """read_test_data(..., shift_zero_digits_images=True) :284-294: one empty image first,
    then all non-empty ones, then the remaining empty ones."""
    empty = [i for i in range(len(digits)) if digits[i] == 0]
    non_empty = [i for i in range(len(digits)) if digits[i] > 0]
    if not empty:
        return images, digits
    order = [empty[0]] + non_empty + empty[1:]
    return images[order], digits[order]


_shift_zero = shift_zero_digits_images        # read_test_data's flag shadows the function name


if __name__ == "__main__":
    parser = argparse.ArgumentParser()                                  # flag surface of multi_mnist.py:312-329
    parser.add_argument("--max-digits", type=int, choices=list(range(7)), default=2)
    parser.add_argument("--max-in-common", type=int, choices=list(range(7)), default=2)
    parser.add_argument("--images-per-digit", type=int, default=20000)
    parser.add_argument("--test-set-size", type=int, default=1000)
    parser.add_argument("--digit-gap", type=int, default=0)
    parser.add_argument("--canvas-margin", type=int, default=0)
    parser.add_argument("--bg-path", default="")
    parser.add_argument("--bg-max-intensity", type=float, default=1.0)
    parser.add_argument("--min-width-scale", type=float, default=1.0)
    parser.add_argument("--max-width-scale", type=float, default=1.0)
    parser.add_argument("--min-height-scale", type=float, default=1.0)
    parser.add_argument("--max-height-scale", type=float, default=1.0)
    parser.add_argument("--min-rotation-angle", type=float, default=0.0)
    parser.add_argument("--max-rotation-angle", type=float, default=0.0)
    parser.add_argument("--use-bounding-box-overlap", action="store_true")
    parser.add_argument("--canvas-size", type=int, default=CANVAS_SIZE)
    parser.add_argument("--tfrecords", action="store_true",
                        help="also write the reference's files: <n>.tfrecords per stratum, common.tfrecords, test.tfrecords")
    args = parser.parse_args()
    os.makedirs(MULTI_MNIST_FOLDER, exist_ok=True)
    bg = read_image(args.bg_path, args.bg_max_intensity) if args.bg_path else None
    strata, rng, source = generate_strata(
        args.max_digits, args.images_per_digit, bg=bg, canvas_dim=args.canvas_size, verbose=True,
        min_w=args.min_width_scale, max_w=args.max_width_scale, min_h=args.min_height_scale, max_h=args.max_height_scale,
        min_ang=args.min_rotation_angle, max_ang=args.max_rotation_angle, gap=args.digit_gap,
        margin=args.canvas_margin, use_pixel_overlap=not args.use_bounding_box_overlap)
    keys = ("images", "indices", "positions", "boxes", "labels", "digits")
    common = {k: [] for k in keys}
    for nd_, st in enumerate(strata):                                   # :347-413
        if args.tfrecords:
            write_to_records(MULTI_MNIST_FOLDER + str(nd_), *[st[k] for k in keys], side=args.canvas_size)
        if nd_ <= args.max_in_common:
            for k in keys:
                common[k].extend(st[k])
    perm = rng.permutation(len(common["images"]))                       # shuffle_lists :215-225
    common = {k: [common[k][i] for i in perm] for k in keys}
    T = args.test_set_size
    if args.tfrecords:
        write_to_records(MULTI_MNIST_FOLDER + "common", *[common[k][T:] for k in keys], side=args.canvas_size)
        write_to_records(MULTI_MNIST_FOLDER + "test", *[common[k][:T] for k in keys], side=args.canvas_size)
    np.savez(MULTI_MNIST_FOLDER + "common.npz", images=np.stack(common["images"][T:]).astype(np.float32),
             digits=np.asarray(common["digits"][T:], np.int32))
    np.savez(MULTI_MNIST_FOLDER + "test.npz", images=np.stack(common["images"][:T]).astype(np.float32),
             digits=np.asarray(common["digits"][:T], np.int32))
    print("glyph source: %s; wrote %d train / %d test images" % (source, len(perm) - T, T))
