"""Driver-run learning gate (BASELINE.json: ">= 98 % digit-count accuracy reproduced"; reference README.md:18: 10 of 10
runs converge towards 98 % over ~25 000 iterations, training.py:100-122 hyper-parameters).

The full evidence (48 seeds x 60 000 iterations per precision, switch-point sweeps, 300-epoch runs) is builder-run and lives
under profiles/ (tools/gate_report.py prints its counts: 39 of 48 bf16 runs and 42 of 48 fp32 runs reach 0.98, 1 and 2 end
below 0.9); these tests put the learning behaviour of the DEFAULT product path -- training.py, bf16 GEMM operands,
backward="reference", hipGraph replays of 50 steps with the in-graph batches -- under `pytest -m gpu`: two seeds at 30 000
iterations (>= 0.95, every count class >= 0.90) and one run of 60 000 iterations (some evaluation >= 0.98, final >= 0.95),
held-out count accuracy on the fixed 1 000-image test set.  Stand-in glyphs: MNIST is not available offline (DESIGN.md
section 2)."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, seed, iterations):
    out = tmp_path / ("run%d_%d" % (seed, iterations))
    cmd = [sys.executable, "training.py", "-r", str(out), "-o", "1", "--iterations", str(iterations), "--print-every", "0",
           "--precision", "bf16", "--seed", str(seed)]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "tf-attend-infer-repeat_amd"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    m = re.search(r"test accuracy ([0-9.]+)\s+test loss ([-0-9.]+)\s+\((\d+) iterations, ([0-9.]+) s\)", r.stdout)
    assert m, r.stdout[-2000:]
    acc, its, wall = float(m.group(1)), int(m.group(3)), float(m.group(4))
    rows = [json.loads(l) for l in open(out / "summary" / "scalars.jsonl")]
    best = max(rw["accuracy"] for rw in rows)
    first = next((rw["step"] for rw in rows if rw["accuracy"] >= 0.98), None)
    print("seed %d: held-out count accuracy %.3f after %d iterations (best %.3f, first evaluation at 0.98: iteration %s, %.1f s)"
          % (seed, acc, its, best, first, wall))
    assert its == iterations
    assert rows[0]["accuracy"] < 0.5                       # it started from chance (count distribution ~ uniform over 0..2)
    return acc, best, first, rows, r.stdout


@pytest.mark.parametrize("seed", [1, 2])
def test_default_training_learns_to_count(tmp_path, seed):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    acc, best, first, rows, _ = _run(tmp_path, seed, 30000)
    # (the lowest final of a run that is not stuck on a count class, over 96 sweep runs of 60 k iterations: 0.955)
    assert acc >= 0.95, (acc, best)
    # per-count accuracies of the last evaluation: every count is learnt, not only the majority one
    last = rows[-1]
    assert min(last["digit_acc_%d_dig" % k] for k in range(3)) >= 0.90, last


def test_default_training_reaches_98_percent(tmp_path):
    """BASELINE.json: ">= 98 % digit-count accuracy reproduced" -- one 60 000-iteration run of the default driver: some
    evaluation on the held-out 1 000 canvases reaches 0.98 and the run ends at or above 0.95 (single evaluations of a
    converged run scatter by ~0.01).  The sweeps behind the README's success rates are under profiles/ (r05_sweep_*,
    r06_gate_*; tools/gate_report.py prints their counts)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    acc, best, first, rows, _ = _run(tmp_path, 3, 60000)
    assert best >= 0.98 and first is not None, (acc, best)
    assert acc >= 0.95, (acc, best)
