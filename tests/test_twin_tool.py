"""tools/twin_train_gpu.py (the torch-autograd twin of the reference used for the clutter / learning comparisons, DESIGN.md
sections 10.7, 11): its hipGraph-replayed iteration is the eager one -- same parameters after a few iterations."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graph_replayed_twin_iteration_equals_the_eager_one(capsys):
    """Same seed, same noise, same batches: the first iteration's ELBO of the graph-replayed run is the eager run's to the
    last bit (the forward has no atomics), and both move the parameters by Adam's first steps.  (The gradients themselves
    are not comparable run to run: at initialisation they ARE the out-of-range residue, and autograd's scatter-adds sum it in
    a different order every time -- this twin's own realisation of the reference's residue.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import twin_train_gpu as tw
    from oracle import air_oracle as ao
    try:
        outs = []
        for extra in ([], ["--no-graph"]):
            p, first_loss = tw.main(["3", "3", "--eval-every", "3"] + extra)
            outs.append(({k: v.detach().clone() for k, v in p.items()}, first_loss))
        init = {k: torch.tensor(v, device="cuda") for k, v in ao.init_params(dict(ao.TRAINING_HP), 3).items()}
    finally:
        torch.set_default_device("cpu")
    moved = [max(float((o[k] - init[k]).abs().max()) for k in o) for o, _ in outs]
    print("twin: first-iteration loss graph %.6f, eager %.6f; largest parameter move after 3 iterations %.2e / %.2e"
          % (outs[0][1], outs[1][1], moved[0], moved[1]))
    assert outs[0][1] == outs[1][1] and outs[0][1] == outs[0][1]          # bit-equal and not NaN
    assert all(1e-4 < m < 1e-3 for m in moved), moved                    # three steps of ~lr = 1e-4 each
