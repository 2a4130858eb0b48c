"""CPU tests of the dataset generator restatement (multi_mnist.py:82-183, 284-294, 341-413)."""
import numpy as np

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
