"""Inference wrapper and attention-window visualisation: counterparts of the
reference's demo/model_wrapper.py and air_model.py:130-157, 211-267."""
import numpy as np
import pytest
import torch

from oracle import air_oracle as ao
from oracle.synth import blob_canvases


def test_resize_bilinear_tf1_matches_definition():
    from air.visualize import resize_bilinear_tf1
    rng = np.random.RandomState(0)
    img = rng.rand(2, 5, 7).astype(np.float32)
    out = resize_bilinear_tf1(torch.tensor(img), 10, 21).numpy()
    assert out.shape == (2, 10, 21)
    for (b, i, j) in [(0, 0, 0), (1, 3, 8), (0, 9, 20), (1, 7, 13)]:
        y, x = i * 5 / 10, j * 7 / 21                      # TF1 resize_bilinear, align_corners=False
        y0, x0 = int(np.floor(y)), int(np.floor(x))
        y1, x1 = min(y0 + 1, 4), min(x0 + 1, 6)
        wy, wx = y - y0, x - x0
        ref = (img[b, y0, x0] * (1 - wx) + img[b, y0, x1] * wx) * (1 - wy) + \
              (img[b, y1, x0] * (1 - wx) + img[b, y1, x1] * wx) * wy
        assert abs(out[b, i, j] - ref) < 1e-6
    # integer zoom of a constant image is the same constant
    assert torch.allclose(resize_bilinear_tf1(torch.full((1, 4, 4), 0.25), 8, 8), torch.full((1, 8, 8), 0.25))


def test_colored_boxes_follow_step_counts():
    from air.visualize import draw_colored_bounding_boxes
    img = torch.full((3, 4, 4), 0.5)
    boxes = torch.zeros(3, 3, 4, 4)
    boxes[:, 0, 0, :] = 1.0          # step 1: top row
    boxes[:, 1, :, 0] = 1.0          # step 2: left column
    steps = torch.tensor([0, 1, 2])
    out = draw_colored_bounding_boxes(img, boxes, steps)
    assert out.shape == (3, 4, 4, 3)
    assert torch.all(out[0] == 0.5)                                    # no step: untouched
    assert torch.all(out[1, 0, :, 0] == 1.0) and torch.all(out[1, 0, :, 1:] == 0.0)   # red top row
    assert torch.all(out[1, 1:, :, :] == 0.5)                           # second box not drawn for 1 step
    assert out[2, 2, 0, 1] == 1.0 and out[2, 2, 0, 0] == 0.0            # green left column for 2 steps


@pytest.mark.gpu
def test_visualize_identity_window_is_canvas_border():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from air.visualize import visualize_reconstructions
    n, C, w, N = 3, 50, 28, 3
    orig = torch.rand(n, C * C, device="cuda")
    rec = torch.rand(n, C * C, device="cuda")
    st = torch.zeros(n, 2, 2, 3, device="cuda")          # T' = 2 < max_steps: padded inside
    st[:, :, 0, 0] = 1.0
    st[:, :, 1, 1] = 1.0                                  # identity: the window covers the canvas
    steps = torch.tensor([0, 1, 2], device="cuda")
    out = visualize_reconstructions(orig, rec, st, steps, C, w, N, zoom=2)
    assert out.shape == (n, 100, 204, 3)
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert torch.all(out[:, :, 100:104, :] == 1.0)        # white stripe
    # image 1 took one step: a red frame along the border of both halves, nothing for image 0
    assert torch.all(out[1, 0, :100, 0] == 1.0) and torch.all(out[1, 0, :100, 1] == 0.0)
    assert torch.all(out[1, 0, 104:, 0] == 1.0)
    from air.visualize import resize_bilinear_tf1
    big = resize_bilinear_tf1(orig.view(n, C, C), 100, 100)
    assert torch.equal(out[0, :, :100, 0], big[0]) and torch.equal(out[0, :, :100, 2], big[0])   # no step: untouched
    assert torch.equal(out[1, 40:60, 40:60, 1], big[1, 40:60, 40:60])                            # interior untouched
    assert not torch.all(out[0, 0, :100, 0] == 1.0)


@pytest.mark.gpu
def test_model_wrapper_matches_model_outputs(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from air import air_model as am
    from demo.model_wrapper import ModelWrapper
    hp = dict(ao.TRAINING_HP)
    B, n = 16, 37
    images, targets = blob_canvases(n, hp["canvas_size"], hp["max_digits"], seed=11)
    am.reset_default_graph()
    model = am.AIRModel(torch.zeros(B, 2500, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"),
                        cnn=False, train=False, **hp)
    model.load_state_dict(ao.init_params(hp, 0))
    noise = ao.make_noise(hp, B, 3)
    model.set_noise(noise)
    model.set_dynamic(z_pres_prior_log_odds=-2.0)
    w = ModelWrapper(model, None, None)
    digits, positions, recs, windows, latents, loss = w.infer([im.reshape(50, 50) for im in images])
    assert len(digits) == len(positions) == len(recs) == len(windows) == len(latents) == len(loss) == n
    # the last chunk is still in the model: compare with its attributes
    last = (n // B) * B
    d = model.rec_num_digits.cpu().numpy()
    for i in range(n - last):
        k = last + i
        assert digits[k] == int(d[i])
        assert positions[k].shape == ((digits[k], 3) if digits[k] else (0,))
        assert windows[k].shape == ((digits[k], 28, 28) if digits[k] else (0,))
        np.testing.assert_array_equal(recs[k], model.reconstruction[i].cpu().numpy().reshape(50, 50))
    # and with the oracle on the first chunk
    o = ao.air_forward(ao.init_params(hp, 0), images[:B], targets[:B], noise, hp, False, -2.0, early_exit=True)
    assert digits[:B] == [int(v) for v in o["rec_num_digits"]]
    for i in range(B):
        np.testing.assert_allclose(recs[i].ravel(), o["reconstruction"][i], atol=2e-5)
        for j in range(digits[i]):
            np.testing.assert_allclose(positions[i][j], [o["rec_scales"][i, j, 0], *o["rec_shifts"][i, j]], atol=5e-5)
