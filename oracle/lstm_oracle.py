"""Float64 NumPy restatement of tf.compat.v1.nn.rnn_cell.LSTMCell (TEST INFRASTRUCTURE ONLY).

The reference builds every encoder / prediction-network layer from this cell
(/root/reference/model.py:57-58, :101-102: LSTMCell(d_model, num_proj=proj_size)) wrapped in
tf.keras.layers.RNN (:62-63, :104-105).  TensorFlow is absent here, so this states the cell's published
arithmetic (TF 1.x rnn_cell_impl.LSTMCell.call, use_peepholes=False, cell_clip=None, proj_clip=None):

    lstm_matrix = concat([x_t, m_{t-1}]) @ kernel + bias            kernel [in + out, 4 n], bias [4 n]
    i, j, f, o  = split(lstm_matrix, 4)                             gate order i, j, f, o
    c_t = sigmoid(f + forget_bias) * c_{t-1} + sigmoid(i) * tanh(j) forget_bias = 1.0 (default)
    m_t = sigmoid(o) * tanh(c_t)
    m_t = m_t @ projection_kernel                                   if num_proj (bias-free)
PARITY STATUS: unpinned (no TensorFlow to run); used to check the weight importer's gate re-ordering."""
from __future__ import annotations

import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def tf1_lstm_cell_sequence(x, kernel, bias, projection_kernel=None, forget_bias=1.0):
    """x [B, T, in] -> outputs [B, T, out] with zero initial state (keras RNN default)."""
    x = np.asarray(x, np.float64)
    kernel = np.asarray(kernel, np.float64)
    bias = np.asarray(bias, np.float64)
    B, T, _ = x.shape
    n = kernel.shape[1] // 4
    out = n if projection_kernel is None else np.asarray(projection_kernel).shape[1]
    c = np.zeros((B, n))
    m = np.zeros((B, out))
    ys = np.zeros((B, T, out))
    for t in range(T):
        z = np.concatenate([x[:, t], m], axis=1) @ kernel + bias
        i, j, f, o = np.split(z, 4, axis=1)
        c = _sigmoid(f + forget_bias) * c + _sigmoid(i) * np.tanh(j)
        m = _sigmoid(o) * np.tanh(c)
        if projection_kernel is not None:
            m = m @ np.asarray(projection_kernel, np.float64)
        ys[:, t] = m
    return ys
