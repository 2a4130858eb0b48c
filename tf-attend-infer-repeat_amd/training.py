"""Training driver -- counterpart of the reference's training.py on the HIP AIRModel.

Same constants (training.py:13-31), same CLI flags (-r/-o/-t, :35-39), same results-folder
layout (:41-61), the same constructor call (:100-122), train + test models sharing variables
(:95-123), periodic evaluation on the fixed 1 000-image test set with the
shift_zero_digits_images ordering (:143-156, 169-200), checkpoints every 10 000 iterations
(:203-207) and the same stdout line (:226).  TensorBoard summaries become JSONL scalars
(summary/scalars.jsonl): loss, accuracy and the per-digit-count breakdown of
AIRModel._summarize_by_digit_count (air_model.py:160-182, 614-617).

The whole dataset lives in HBM.  A batch comes out of tf.train.shuffle_batch's queue, kept on the
device (multi_mnist.ShuffleBatchQueue; multi_mnist.py:240-249: capacity 10 000 + 10 * batch, min_after_dequeue
10 000, over the epoch-repeating record stream of training.py:76-81) and is gathered into the train model's input
buffer; with --print-every 0 fifty train steps and their batches are one hipGraph replay (the queue's picks for the
fifty batches in its first launch, a row gather in front of every step), and the evaluation every 50 iterations is one replay of the
test model's forward + one summaries launch whose numbers are fetched without blocking (SummaryWriter).
"""
import argparse
import json
import os
import shutil
import time

import numpy as np
import torch

from multi_mnist import ShuffleBatchQueue, generate_dataset, shift_zero_digits_images
from air.air_model import AIRModel

EPOCHS = 300
BATCH_SIZE = 64
CANVAS_SIZE = 50

NUM_SUMMARIES_EACH_ITERATIONS = 50
VAR_SUMMARIES_EACH_ITERATIONS = 250
IMG_SUMMARIES_EACH_ITERATIONS = 500
GRAD_SUMMARIES_EACH_ITERATIONS = 100
SAVE_PARAMS_EACH_ITERATIONS = 10000
NUM_IMAGES_TO_SAVE = 60

MIN_AFTER_DEQUEUE = 10000                 # multi_mnist.py:246-247 (capacity = this + 10 * batch)
DEFAULT_READER_THREADS = 4
DEFAULT_RESULTS_FOLDER = "air_results"
TRAIN_DATA_FILE = "multi_mnist_data/common.npz"
TEST_DATA_FILE = "multi_mnist_data/test.npz"


def load_data(bg_path="", bg_max_intensity=1.0):
    if os.path.exists(TRAIN_DATA_FILE) and os.path.exists(TEST_DATA_FILE):
        tr, te = np.load(TRAIN_DATA_FILE), np.load(TEST_DATA_FILE)
        return tr["images"], tr["digits"], te["images"], te["digits"]
    rec_tr, rec_te = TRAIN_DATA_FILE.replace(".npz", ".tfrecords"), TEST_DATA_FILE.replace(".npz", ".tfrecords")
    if os.path.exists(rec_tr) and os.path.exists(rec_te):
        # the reference's own files (training.py:23-24: multi_mnist_data/common.tfrecords, test.tfrecords)
        from multi_mnist import read_test_data
        print("Reading the reference's TFRecord files...")
        tri, trd, *_ = read_test_data(rec_tr)
        tei, ted, *_ = read_test_data(rec_te)
        return tri, trd.astype(np.int32), tei, ted.astype(np.int32)
    print("Generating multi-digit dataset in memory (multi_mnist.py defaults)...")
    bg = None
    if bg_path:                                            # clutter (multi_mnist.py --bg-path / --bg-max-intensity)
        if ".npz:" in bg_path:                             # "<file>.npz:<key>": an already decoded background
            f, key = bg_path.rsplit(":", 1)
            bg = np.load(f)[key].astype(np.float32)
            if bg.max() > 0:
                bg = bg / bg.max() * min(bg_max_intensity, 1.0)
        else:
            from multi_mnist import read_image
            bg = read_image(bg_path, bg_max_intensity)
    ds = generate_dataset(bg=bg)
    return ds["train_images"], ds["train_digits"], ds["test_images"], ds["test_digits"]


class SummaryWriter:
    """The reference's numeric summaries (air_model.py:160-209, 608-625; training.py:171-180 evaluates them on the test
    model every NUM_SUMMARIES_EACH_ITERATIONS) as rows of summary/scalars.jsonl.  AIRModel.numeric_summaries() is one
    launch into a device vector; the vector is copied into a pinned host ring without blocking and a row is written
    when the NEXT evaluation has been enqueued, so the host never waits for the step it has just launched."""

    def __init__(self, model, path, t0, slots=2):
        self.model, self.t0 = model, t0
        self.names = model.summary_names()
        self.dev = torch.empty(len(self.names), dtype=torch.float32, device=model.input_images.device)
        self.host = torch.empty(slots, len(self.names), dtype=torch.float32).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(slots)]
        self.pending, self.k, self.file = [], 0, open(path, "w")

    def push(self, step):
        slot = self.k % len(self.events)
        self.k += 1
        self.model.numeric_summaries(self.dev)
        self.host[slot].copy_(self.dev, non_blocking=True)
        self.events[slot].record()
        self.pending.append((slot, step))
        while len(self.pending) > 1:
            self._drain_one()

    def _drain_one(self):
        slot, step = self.pending.pop(0)
        self.events[slot].synchronize()
        row = {"step": step, "wall_s": round(time.perf_counter() - self.t0, 3)}
        row.update({k: (round(v, 5) if v == v else None) for k, v in zip(self.names, self.host[slot].tolist())})
        self.file.write(json.dumps(row) + "\n")
        self.file.flush()

    def flush(self):
        while self.pending:
            self._drain_one()


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-r", "--results-folder", default=DEFAULT_RESULTS_FOLDER)
    parser.add_argument("-o", "--overwrite-results", type=int, choices=[0, 1], default=0)
    parser.add_argument("-t", "--reader-threads", type=int, default=DEFAULT_READER_THREADS)   # kept for CLI parity
    parser.add_argument("--iterations", type=int, default=0, help="stop after this many iterations (0 = EPOCHS)")
    parser.add_argument("--print-every", type=int, default=1)
    parser.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    parser.add_argument("--no-graph", action="store_true")
    parser.add_argument("--tf-checkpoints", action="store_true",
                        help="also write TensorFlow bundles models/air-model-<step>.{index,data-00000-of-00001} (tf.train.Saver layout)")
    parser.add_argument("--bg-path", default="", help="clutter background for the in-memory dataset (png, or file.npz:key)")
    parser.add_argument("--bg-max-intensity", type=float, default=1.0)
    parser.add_argument("--graph-steps", type=int, default=50,
                        help="train steps per hipGraph replay when --print-every 0 (a divisor of 50)")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--backward", default="reference", choices=["reference", "reference_carried", "exact"],
                        help="sampler backward: the reference graph's op order, the same streams with the long ones in 16 "
                             "carried chunks per tap, or the exact adjoint")
    parser.add_argument("--late-backward", default="", choices=["", "reference", "reference_carried", "exact"],
                        help="switch the sampler backward to this order from iteration --late-backward-from on "
                             "(AIRModel(backward=(first, late, iteration)))")
    parser.add_argument("--late-backward-from", type=int, default=5000,
                        help="(opt-in: over 48 seeds per precision no switch point keeps the reference order's success rate within 60 000 "
                             "iterations; over the full 276 300 the two end alike and this saves ~3 s -- DESIGN.md section 11.1)")
    args = parser.parse_args()

    # results folder handling, training.py:41-61
    if os.path.exists(args.results_folder):
        if args.overwrite_results:
            shutil.rmtree(args.results_folder, ignore_errors=True)
        else:
            folder, i = args.results_folder, 0
            args.results_folder = "{}_{}".format(folder, i)
            while os.path.exists(args.results_folder):
                i += 1
                args.results_folder = "{}_{}".format(folder, i)
    models_folder = args.results_folder + "/models/"
    summaries_folder = args.results_folder + "/summary/"
    for d in (args.results_folder, models_folder, summaries_folder):
        os.makedirs(d)

    dev = torch.device("cuda", 0)
    print("Creating input pipeline...")
    tr_im, tr_dg, te_im, te_dg = load_data(args.bg_path, args.bg_max_intensity)
    te_im, te_dg = shift_zero_digits_images(te_im, te_dg)
    train_images = torch.tensor(tr_im, device=dev)
    train_digits = torch.tensor(tr_dg.astype(np.int32), device=dev)
    test_data = torch.tensor(np.ascontiguousarray(te_im), device=dev)
    test_targets = torch.tensor(np.ascontiguousarray(te_dg).astype(np.int32), device=dev)
    train_data = torch.zeros(BATCH_SIZE, CANVAS_SIZE ** 2, device=dev)
    train_targets = torch.zeros(BATCH_SIZE, dtype=torch.int32, device=dev)

    backward = args.backward
    if args.late_backward and args.late_backward != args.backward:
        if args.late_backward_from < 0:
            parser.error("--late-backward-from must be >= 0")
        backward = (args.backward, args.late_backward, args.late_backward_from)
    models = []
    model_inputs = [[train_data, train_targets], [test_data, test_targets]]
    for i in range(2):
        print("Creating {0} model...".format("training" if i == 0 else "testing"))
        models.append(
            AIRModel(
                model_inputs[i][0], model_inputs[i][1],
                max_steps=3, max_digits=2, rnn_units=256, canvas_size=CANVAS_SIZE, windows_size=28,
                vae_latent_dimensions=50, vae_recognition_units=(512, 256), vae_generative_units=(256, 512),
                scale_prior_mean=-1.0, scale_prior_variance=0.05, shift_prior_mean=0.0, shift_prior_variance=1.0,
                vae_prior_mean=0.0, vae_prior_variance=1.0, vae_likelihood_std=0.3,
                scale_hidden_units=64, shift_hidden_units=64, z_pres_hidden_units=64,
                z_pres_prior_log_odds=-0.01, z_pres_temperature=1.0, stopping_threshold=0.99,
                learning_rate=1e-4, gradient_clipping_norm=1.0, cnn=False, cnn_filters=8,
                num_summary_images=NUM_IMAGES_TO_SAVE, train=(i == 0), reuse=(i == 1), scope="air",
                annealing_schedules={
                    "z_pres_prior_log_odds": {
                        "init": 10000.0, "min": 0.000000001,
                        "factor": 0.1, "iters": 3000,
                        "staircase": False, "log": True
                    },
                },
                seed=args.seed, gemm_precision=args.precision, backward=backward,
            )
        )
    train_model, test_model = models
    n_train = train_images.shape[0]
    total = args.iterations if args.iterations > 0 else (n_train // BATCH_SIZE) * EPOCHS

    # The input queue is device work too (read_and_decode, multi_mnist.py:228-249): the RandomShuffleQueue's resident
    # record indices and the stream position live in HBM.  When nothing is printed per step, --graph-steps train steps
    # are captured per hipGraph replay together with their batches: one launch at the head of the replay makes the picks of
    # all its batches, one row gather sits in front of every step.
    batches = ShuffleBatchQueue(train_images, train_digits, BATCH_SIZE, train_data, train_targets,
                                seed=0x5348554646 + args.seed, min_after_dequeue=MIN_AFTER_DEQUEUE)
    gsteps = 1
    if not args.no_graph:
        if args.print_every == 0 and NUM_SUMMARIES_EACH_ITERATIONS % args.graph_steps == 0 and args.graph_steps > 1:
            gsteps = args.graph_steps

    def capture():
        if args.no_graph:
            return
        if gsteps > 1:
            between, after = batches.graph_hooks(gsteps)
            train_model.capture_graph(steps=gsteps, between_steps=between, after_steps=after)
        else:
            train_model.capture_graph(steps=1)
    capture()
    if not args.no_graph:
        test_model.capture_graph()                                   # the evaluation pass: one replay

    print("Training...")
    print()
    step = 0
    order = train_model.backward
    t0 = time.perf_counter()
    writer = SummaryWriter(test_model, summaries_folder + "scalars.jsonl", t0)
    while step < total:
        if step % NUM_SUMMARIES_EACH_ITERATIONS == 0:
            test_model.forward()
            writer.push(step)
        if step % SAVE_PARAMS_EACH_ITERATIONS == 0:
            torch.save(train_model.state_dict(), models_folder + "air-model-%d.pt" % step)
            if args.tf_checkpoints:                                   # training.py:203-207 saver.save(..., global_step)
                train_model.save_tf_checkpoint(models_folder + "air-model-%d" % step)
        if train_model.backward != order:                            # (a backward schedule: the model switched by itself)
            order = train_model.backward
            print("iteration {}: sampler backward order -> {}".format(step - gsteps, order))
        if gsteps == 1:
            batches.next_batch()
        train_model.training()
        step += gsteps
        if args.print_every and step % args.print_every == 0:
            print("iteration {}\tloss {:.3f}\taccuracy {:.2f}".format(
                int(train_model.global_step), float(train_model.loss), float(train_model.accuracy)))
    torch.cuda.synchronize()
    writer.flush()
    test_model.forward()
    print()
    print("training has ended")
    print("test accuracy {:.4f}  test loss {:.3f}  ({} iterations, {:.1f} s)".format(
        float(test_model.accuracy), float(test_model.loss), step, time.perf_counter() - t0))
    torch.save(train_model.state_dict(), models_folder + "air-model-%d.pt" % step)
    if args.tf_checkpoints:
        train_model.save_tf_checkpoint(models_folder + "air-model-%d" % step)


if __name__ == "__main__":
    main()
