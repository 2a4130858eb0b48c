"""Training driver -- counterpart of the reference's training.py on the HIP AIRModel.

Same constants (training.py:13-31), same CLI flags (-r/-o/-t, :35-39), same results-folder
layout (:41-61), the same constructor call (:100-122), train + test models sharing variables
(:95-123), periodic evaluation on the fixed 1 000-image test set with the
shift_zero_digits_images ordering (:143-156, 169-200), checkpoints every 10 000 iterations
(:203-207) and the same stdout line (:226).  TensorBoard summaries become JSONL scalars
(summary/scalars.jsonl): loss, accuracy and the per-digit-count breakdown of
AIRModel._summarize_by_digit_count (air_model.py:160-182, 614-617).

The whole dataset lives in HBM.  A batch comes out of tf.train.shuffle_batch's queue, kept on the
device (multi_mnist.py:240-249: capacity 10 000 + 10 * batch, min_after_dequeue 10 000, over the epoch-repeating
record stream of training.py:76-81; air_shuffle_batch_dequeue, include/air_hip.h) and is gathered into the
train model's input buffer; queue, gather and train step are captured together in the hipGraph replay.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import time

import numpy as np
import torch

from air import _hip as H
from multi_mnist import generate_dataset, shift_zero_digits_images
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


class Summaries:
    """Scalar summaries of the reference (air_model.py:160-209, 614-632), collected as device
    scalars and fetched with ONE device->host copy per evaluation."""

    def __init__(self, targets, max_digits, max_steps):
        self.targets, self.max_digits, self.max_steps = targets, max_digits, max_steps
        self.names, self.vals = [], []

    def _mean(self, v, mask):
        m = mask.float()
        return (v * m).sum() / m.sum()                              # NaN when the group is empty, as tf.reduce_mean

    def by_digit_count(self, name, values, mask=None):
        """_summarize_by_digit_count :160-182"""
        v = values.float()
        ok = torch.ones_like(v, dtype=torch.bool) if mask is None else mask
        for i in range(self.max_digits + 1):
            self.names.append("%s_%d_dig" % (name, i))
            self.vals.append(self._mean(v, ok & (self.targets == i)))
        self.names.append(name + "_all_dig")
        self.vals.append(self._mean(v, ok))

    def by_step(self, tensor, steps, name, one_more_step=False, all_steps=False):
        """_summarize_by_step :184-209 (tensor [B, T'] is padded to max_steps columns)"""
        T = tensor.shape[1]
        for i in range(self.max_steps):
            col = tensor[:, i] if i < T else torch.zeros_like(tensor[:, 0])
            mask = None if all_steps else steps > (i - (1 if one_more_step else 0))
            self.by_digit_count("%s_%d_step" % (name, i + 1), col, mask)

    def fetch(self):
        vals = torch.stack(self.vals).cpu().tolist()
        return dict(zip(self.names, vals))


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
    parser.add_argument("--graph-steps", type=int, default=10, help="train steps per hipGraph replay when --print-every 0")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--backward", default="reference", choices=["reference", "reference_carried", "reference_blocked", "taps", "exact"],
                        help="sampler backward: the reference graph's op order (default), the same streams in 16 chunks per tap, "
                             "per-tap sums, or the exact adjoint")
    parser.add_argument("--late-backward", default="", choices=["", "reference", "reference_carried", "reference_blocked", "taps", "exact"],
                        help="switch the sampler backward to this order from iteration --late-backward-from on")
    parser.add_argument("--late-backward-from", type=int, default=5000,
                        help="(a multiple of 50) by then ink is explained and the out-of-range residue no longer rules the gradient")
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
                seed=args.seed, gemm_precision=args.precision, backward=args.backward,
            )
        )
    train_model, test_model = models
    n_train = train_images.shape[0]
    total = args.iterations if args.iterations > 0 else (n_train // BATCH_SIZE) * EPOCHS
    scalars = open(summaries_folder + "scalars.jsonl", "w")

    # The input queue is device work too (read_and_decode, multi_mnist.py:228-249): the RandomShuffleQueue's resident
    # record indices and the stream position live in HBM, so several train steps (dequeue + gather + step) are captured
    # per hipGraph replay when nothing is printed per step.
    queue = torch.zeros(MIN_AFTER_DEQUEUE + 10 * BATCH_SIZE, dtype=torch.int32, device=dev)
    queue_state = torch.zeros(2, dtype=torch.int64, device=dev)
    picks = torch.zeros(BATCH_SIZE, dtype=torch.int32, device=dev)
    sq = H.ShuffleBatch(queue.data_ptr(), queue_state.data_ptr(), picks.data_ptr(), queue.numel(), BATCH_SIZE,
                        MIN_AFTER_DEQUEUE, n_train, 0x5348554646 + args.seed)

    def _s():
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def next_batch(_i=0):
        H.check(H.lib().air_shuffle_batch_dequeue(C.byref(sq), _s()), "air_shuffle_batch_dequeue")
        torch.index_select(train_images, 0, picks, out=train_data)
        torch.index_select(train_digits, 0, picks, out=train_targets)

    gsteps = 1
    if not args.no_graph:
        if args.print_every == 0 and NUM_SUMMARIES_EACH_ITERATIONS % args.graph_steps == 0:
            gsteps = args.graph_steps

    def capture():
        if not args.no_graph:
            train_model.capture_graph(steps=gsteps, between_steps=next_batch if gsteps > 1 else None)
    capture()
    H.check(H.lib().air_shuffle_batch_init(C.byref(sq), _s()), "air_shuffle_batch_init")   # (after the capture: nothing consumed)

    print("Training...")
    print()
    step = 0
    t0 = time.perf_counter()
    while step < total:
        if step % NUM_SUMMARIES_EACH_ITERATIONS == 0:
            test_model.forward()
            digs = test_model.rec_num_digits
            sm = Summaries(test_targets, 2, 3)
            sm.names += ["loss", "accuracy"]
            sm.vals += [test_model.loss.float(), test_model.accuracy.float()]
            sm.by_digit_count("steps", digs)                                        # :614-617
            sm.by_digit_count("rec_loss", test_model.reconstruction_loss)
            sm.by_digit_count("digit_acc", digs == test_targets)
            sm.by_digit_count("total_loss", test_model.loss_per_item)
            sm.by_step(test_model.rec_scales[:, :, 0], digs, "scale")               # :620-625
            sm.by_step(test_model.z_pres_probs, digs, "z_pres_prob", all_steps=True)
            sm.by_step(test_model.z_pres_kls, digs, "z_pres_kl", one_more_step=True)
            sm.by_step(test_model.scale_kls, digs, "scale_kl")
            sm.by_step(test_model.shift_kls, digs, "shift_kl")
            sm.by_step(test_model.vae_kls, digs, "vae_kl")
            row = {"step": step, "wall_s": round(time.perf_counter() - t0, 3)}
            row.update({k: (round(v, 5) if v == v else None) for k, v in sm.fetch().items()})
            scalars.write(json.dumps(row) + "\n")
            scalars.flush()
        if step % SAVE_PARAMS_EACH_ITERATIONS == 0:
            torch.save(train_model.state_dict(), models_folder + "air-model-%d.pt" % step)
            if args.tf_checkpoints:                                   # training.py:203-207 saver.save(..., global_step)
                train_model.save_tf_checkpoint(models_folder + "air-model-%d" % step)
        if args.late_backward and step == args.late_backward_from:
            train_model.set_backward(args.late_backward)            # launch lists rebuilt; the graph is captured again
            capture()
        if gsteps == 1:
            next_batch()
        train_model.training()
        step += gsteps
        if args.print_every and step % args.print_every == 0:
            print("iteration {}\tloss {:.3f}\taccuracy {:.2f}".format(
                int(train_model.global_step), float(train_model.loss), float(train_model.accuracy)))
    torch.cuda.synchronize()
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
