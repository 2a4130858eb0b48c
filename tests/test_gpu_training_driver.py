"""End-to-end run of the training driver on the GPU (counterpart of /root/reference/training.py):
CLI flags (:35-39), results-folder layout (:41-61), evaluation cadence (:20-24, 169-200), the stdout
line (:226), checkpoints at step 0 / every 10 000 iterations / at the end (:203-207) and the scalar
summaries of air_model.py:160-182, 614-625 (as JSONL).  Runs training.py in a child process."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")


@pytest.mark.parametrize("print_every,precision", [(50, "fp32"), (0, "bf16")])
def test_training_driver_end_to_end(tmp_path, print_every, precision):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = str(tmp_path / "air_results")
    cmd = [sys.executable, "training.py", "-r", res, "-o", "1", "-t", "4", "--iterations", "300",
           "--print-every", str(print_every), "--precision", precision, "--tf-checkpoints"]
    p = subprocess.run(cmd, cwd=PKG, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    assert "Creating training model..." in out and "Creating testing model..." in out and "Training..." in out
    lines = re.findall(r"^iteration (\d+)\tloss (-?\d+\.\d{3})\taccuracy (\d\.\d{2})$", out, flags=re.M)   # training.py:226
    if print_every:
        assert [int(l[0]) for l in lines] == list(range(50, 301, 50))
        assert all(np.isfinite(float(l[1])) and 0.0 <= float(l[2]) <= 1.0 for l in lines)
    else:
        assert not lines
    m = re.search(r"test accuracy (\d\.\d+)  test loss (-?\d+\.\d+)  \(300 iterations", out)
    assert m and np.isfinite(float(m.group(2)))
    # results layout: models/ and summary/ under the results folder
    models = sorted(os.listdir(os.path.join(res, "models")))
    assert "air-model-0.pt" in models and "air-model-300.pt" in models            # step 0 and the final step
    assert not any(re.fullmatch(r"air-model-(50|100|150|200|250)\.pt", f) for f in models)   # cadence is 10 000
    assert "air-model-0.index" in models and "air-model-0.data-00000-of-00001" in models      # tf.train.Saver layout
    rows = [json.loads(l) for l in open(os.path.join(res, "summary", "scalars.jsonl"))]
    assert [r["step"] for r in rows] == list(range(0, 300, 50))                   # NUM_SUMMARIES_EACH_ITERATIONS
    keys = set(rows[0])
    for k in ("loss", "accuracy", "wall_s"):
        assert k in keys
    # per-digit-count breakdown (air_model.py:160-182): all / 0 / 1 / 2 digits
    for stem in ("steps", "rec_loss", "digit_acc", "total_loss"):
        assert any(k.startswith(stem) for k in keys), (stem, sorted(keys))
    assert all(0.0 <= r["accuracy"] <= 1.0 and np.isfinite(r["loss"]) for r in rows)
    # the saved state resumes: global_step and TF-named variables
    sd = torch.load(os.path.join(res, "models", "air-model-300.pt"))
    assert int(sd["global_step"]) == 300 and tuple(sd["rnn/kernel"].shape) == (2756, 1024)
    # a second run without -o 1 must not overwrite: it gets the next free folder name (:47-56)
    p2 = subprocess.run([sys.executable, "training.py", "-r", res, "--iterations", "50", "--print-every", "0",
                         "--precision", precision], cwd=PKG, capture_output=True, text=True, timeout=900)
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert os.path.isdir(res + "_0") and os.path.exists(os.path.join(res, "models", "air-model-300.pt"))


def test_training_driver_backward_schedule(tmp_path):
    """--late-backward / --late-backward-from: AIRModel(backward=(first, late, N)) -- the driver reports the switch at its
    iteration and the run goes on (the schedule is opt-in: DESIGN.md section 11.1)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = str(tmp_path / "air_results")
    cmd = [sys.executable, "training.py", "-r", res, "-o", "1", "--iterations", "300", "--print-every", "0", "--precision", "bf16",
           "--late-backward", "reference_carried", "--late-backward-from", "100"]
    p = subprocess.run(cmd, cwd=PKG, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "iteration 100: sampler backward order -> reference_carried" in p.stdout, p.stdout[-1500:]
    rows = [json.loads(l) for l in open(os.path.join(res, "summary", "scalars.jsonl"))]
    assert [r["step"] for r in rows] == list(range(0, 300, 50)) and all(np.isfinite(r["loss"]) for r in rows)
    assert len(rows[0]) == 92                                                     # step, wall_s + the reference's 90 numeric summaries
