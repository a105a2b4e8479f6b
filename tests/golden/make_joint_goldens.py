"""Generates tests/golden/joint/*.npz: inputs and expected outputs of the fused joint + loss path from the float64
oracle (oracle/rnnt_oracle.py: joint_loss_and_grads for the f32 joint, joint_loss_and_grads_f16 for the f16-MFMA joint
with its stated binary16 roundings).  Data only; the reference itself cannot run here (TensorFlow + warp-transducer).
    python tests/golden/make_joint_goldens.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import rnnt_oracle as orc  # noqa: E402

CASES = {
    # name: (B, T, U, H, J, V, dtype)
    "joint_f32_char": (3, 14, 9, 12, 64, 28, "f32"),
    "joint_f16_v512": (2, 11, 7, 12, 128, 512, "f16"),
}


def make(name, B, T, U, H, J, V, dtype, seed=4321):
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    enc = rng.normal(size=(B, T, H)).astype(np.float32)
    pred = rng.normal(size=(B, U, H)).astype(np.float32)
    lim1, lim2 = np.sqrt(6.0 / (H + J)), np.sqrt(6.0 / (J + V))
    W1 = rng.uniform(-lim1, lim1, size=(H, J)).astype(np.float32)
    b1 = (0.1 * rng.normal(size=J)).astype(np.float32)
    W2 = (3.0 * rng.uniform(-lim2, lim2, size=(J, V))).astype(np.float32)
    b2 = (0.1 * rng.normal(size=V)).astype(np.float32)
    labels = rng.integers(1, V, size=(B, U - 1)).astype(np.int32)
    il = rng.integers((T + 1) // 2, T + 1, size=B).astype(np.int32)
    ll = rng.integers(U // 2, U, size=B).astype(np.int32)
    il[0], ll[0] = T, U - 1
    scale = np.linspace(0.5, 1.5, B)
    fn = orc.joint_loss_and_grads if dtype == "f32" else orc.joint_loss_and_grads_f16
    r = fn(enc, pred, W1, b1, W2, b2, labels, il, ll, cost_scale=scale)
    np.savez_compressed(
        os.path.join(HERE, "joint", name + ".npz"), enc=enc, pred=pred, W1=W1, b1=b1, W2=W2, b2=b2, labels=labels,
        input_lengths=il, label_lengths=ll, cost_scale=scale, joint_dtype=np.array(dtype), costs=r["costs"],
        **{k: r[k].astype(np.float32) for k in ("d_enc", "d_pred", "dW1", "db1", "dW2", "db2")})
    print(name, r["costs"])


if __name__ == "__main__":
    for k, v in CASES.items():
        make(k, *v)
