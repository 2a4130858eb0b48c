"""Headless counterpart of the reference's demo.py: restores a trained model, runs inference on a
set of canvases and writes what the tkinter demo / the TensorBoard image summary would show --
one PNG grid of [original + attention boxes | reconstruction + attention boxes] (like
images/rec_samples.png) and a JSON with the inferred digit counts and [s, x, y] positions.

  python demo.py --model air_results/models/air-model-60000.pt [--data multi_mnist_data/test.npz]
                 [--num-images 60] [--out demo_out]

The interactive tkinter window (demo/demo_window.py, pixel_canvas.py) is out of scope.
"""
import argparse
import json
import os

import numpy as np
import torch

from air.air_model import AIRModel
from air.visualize import save_image_grid, visualize_reconstructions
from demo.model_wrapper import ModelWrapper

CANVAS_SIZE = 50
WINDOW_SIZE = 28
MODEL_PATH = "./air_results/models/air-model-0.pt"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=MODEL_PATH)
    ap.add_argument("--data", default="", help=".npz with `images` [n,2500] (default: a freshly generated test set)")
    ap.add_argument("--num-images", type=int, default=60)
    ap.add_argument("--out", default="demo_out")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    args = ap.parse_args()

    dev = torch.device("cuda", 0)
    if args.data:
        d = np.load(args.data)
        images = d["images"][:args.num_images].astype(np.float32)
        targets = d["digits"][:args.num_images].astype(np.int32) if "digits" in d else np.zeros(len(images), np.int32)
    else:
        from multi_mnist import generate_dataset
        ds = generate_dataset(images_per_digit=max(args.num_images, 30), test_set_size=args.num_images, seed=1)
        images, targets = ds["test_images"], ds["test_digits"]
    n = len(images)

    test_data = torch.zeros(n, CANVAS_SIZE ** 2, device=dev)
    test_targets = torch.tensor(targets, dtype=torch.int32, device=dev)
    print("Creating model...")
    air_model = AIRModel(                                                           # demo.py:19-27
        test_data, test_targets,
        max_steps=3, rnn_units=256, canvas_size=CANVAS_SIZE, windows_size=WINDOW_SIZE,
        vae_latent_dimensions=50, vae_recognition_units=(512, 256), vae_generative_units=(256, 512),
        vae_likelihood_std=0.3, scale_hidden_units=64, shift_hidden_units=64, z_pres_hidden_units=64,
        z_pres_temperature=1.0, stopping_threshold=0.99, cnn=False,
        train=False, reuse=False, scope="air", gemm_precision=args.precision,
    )
    print("Restoring model...")
    if os.path.exists(args.model + ".index"):                 # a TensorFlow bundle (the reference's model/air-model)
        air_model.load_tf_checkpoint(args.model)
    else:
        air_model.load_state_dict(torch.load(args.model, map_location="cpu"), load_optimizer=False)   # inference: variables only
    wrapper = ModelWrapper(air_model, None, test_data, CANVAS_SIZE, WINDOW_SIZE)

    digits, positions, recs, windows, latents, loss = wrapper.infer(list(images))
    os.makedirs(args.out, exist_ok=True)
    vis = visualize_reconstructions(air_model.input_images, air_model.reconstruction, air_model.rec_st_back,
                                    air_model.rec_num_digits, CANVAS_SIZE, WINDOW_SIZE, air_model.max_steps, zoom=2)
    png = save_image_grid(vis, os.path.join(args.out, "rec_samples.png"), columns=10)
    rows = [{"image": i, "target_digits": int(targets[i]), "inferred_digits": int(digits[i]),
             "positions_s_x_y": [[round(float(v), 4) for v in p] for p in positions[i]],
             "reconstruction_loss": round(float(loss[i]), 3)} for i in range(n)]
    with open(os.path.join(args.out, "inference.json"), "w") as f:
        json.dump(rows, f, indent=1)
    acc = float(np.mean([r["target_digits"] == r["inferred_digits"] for r in rows]))
    print("wrote %s and inference.json (%d images, digit-count accuracy %.3f)" % (png, n, acc))


if __name__ == "__main__":
    main()
