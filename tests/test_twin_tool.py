"""tools/twin_train_gpu.py (the torch-autograd twin of the reference used for the clutter / learning comparisons, DESIGN.md
sections 10.7, 11): its hipGraph-replayed iteration is the eager one -- same parameters after a few iterations."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graph_replayed_twin_iteration_equals_the_eager_one(capsys):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import twin_train_gpu as tw
    try:
        outs = []
        for extra in ([], ["--no-graph"]):
            p = tw.main(["3", "6", "--eval-every", "4"] + extra)
            outs.append({k: v.detach().clone() for k, v in p.items()})
    finally:
        torch.set_default_device("cpu")
    worst = max(float((outs[0][k] - outs[1][k]).abs().max()) for k in outs[0])
    moved = max(float((outs[0][k] - torch.tensor(__import__("oracle.air_oracle", fromlist=["x"]).init_params(
        dict(__import__("oracle.air_oracle", fromlist=["x"]).TRAINING_HP), 3)[k], device="cuda")).abs().max()) for k in outs[0])
    print("twin: graph vs eager max |d param| %.3e after 6 iterations (parameters moved by %.3e)" % (worst, moved))
    assert moved > 1e-4
    assert worst == 0.0
