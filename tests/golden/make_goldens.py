"""Generates tests/golden/*.npz from the float64 oracle (oracle/rnnt_oracle.py).

The reference cannot be imported in any container (it needs TensorFlow and the un-vendored
warp-transducer op), so these fixtures are outputs of the build's own oracle, itself pinned by
tests/golden/kat_small.json and finite differences (tests/test_oracle.py).  Re-run with
    python tests/golden/make_goldens.py
Inputs are seeded; the files are data only (inputs + expected outputs)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import rnnt_oracle as orc  # noqa: E402

CASES = {
    # name: (B, T, U, V, ragged, blank)
    "c1_plumbing": (4, 50, 20, 28, False, 0),      # BASELINE.json configs[0]
    "ragged_small": (5, 23, 9, 28, True, 0),
    "charvocab31": (3, 12, 7, 31, True, 0),        # utils/vocabulary.py: 31 symbols, V % 4 != 0
    "bigv_wavepath": (2, 6, 4, 300, True, 0),
    "blank_last": (2, 9, 5, 12, True, 11),
}


def make(name, B, T, U, V, ragged, blank, seed=1234):
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    acts = rng.normal(size=(B, T, U, V)).astype(np.float32)
    lab_pool = [v for v in range(V) if v != blank]
    labels = rng.choice(lab_pool, size=(B, max(U - 1, 1))).astype(np.int32)[:, : max(U - 1, 0)]
    if ragged:
        il = rng.integers((T + 1) // 2, T + 1, size=B).astype(np.int32)
        ll = rng.integers(U // 2, U, size=B).astype(np.int32)
        il[0], ll[0] = T, U - 1            # one full-length utterance
        if B > 2:
            ll[1] = 0                      # an empty transcript
            il[2] = 1                      # a single frame
    else:
        il = np.full(B, T, np.int32)
        ll = np.full(B, U - 1, np.int32)
    costs, grads = orc.rnnt_loss_and_grad(acts, labels, il, ll, blank=blank, fused_softmax=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), acts=acts, labels=labels, input_lengths=il, label_lengths=ll,
        blank=np.int32(blank), costs=costs, grads=grads.astype(np.float32))
    print(name, acts.shape, costs)


if __name__ == "__main__":
    for k, v in CASES.items():
        make(k, *v)
