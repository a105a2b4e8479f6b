"""Accuracy / WER bookkeeping of the eval step (reference utils/metrics.py:6-92, used at
run_rnnt.py:380-441).  Host-side integer work on one utterance at a time -- plain Python.

Semantics kept from the reference:
  * error_rate = Levenshtein distance (insert/delete/substitute, cost 1, NOT normalised: tf.edit_distance(...,
    normalize=False), :22) divided by max(len(y_true), len(decoded)) measured on the DENSE shapes (:8-11,23);
  * id sequences go through tf.sparse.from_dense (:16,21), which drops zeros: padding / blank ids do not take part
    in the distance (but they do count in the dense lengths of the denominator);
  * string (token) sequences keep every element (string_to_sparse, :28-39);
  * Accuracy = 1 - error_rate on ids of the first utterance, decode capped at len(y_true) (:62-77);
  * WER = the same rate over the space-separated tokens of the two texts (:42-59, 80-92)."""
from __future__ import annotations

from typing import Callable, Sequence


def edit_distance(hyp: Sequence, truth: Sequence) -> int:
    """Levenshtein distance between two sequences (what tf.edit_distance computes with normalize=False)."""
    n, m = len(hyp), len(truth)
    if n == 0:
        return m
    prev = list(range(m + 1))
    for i in range(1, n + 1):
        cur = [i] + [0] * m
        hi = hyp[i - 1]
        for j in range(1, m + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (hi != truth[j - 1]))
        prev = cur
    return prev[m]


def _as_list(x) -> list:
    if hasattr(x, "tolist"):
        x = x.tolist()
    x = list(x)
    while len(x) == 1 and isinstance(x[0], (list, tuple)):  # [1, n] -> [n]
        x = list(x[0])
    return x


def error_rate(y_true, decoded) -> float:
    """utils/metrics.py:6-25.  Integer sequences lose their zeros before the distance; strings keep everything."""
    t, d = _as_list(y_true), _as_list(decoded)
    max_length = max(len(t), len(d))
    if max_length == 0:
        return 0.0
    is_str = any(isinstance(v, str) for v in t + d)
    if not is_str:
        t_eff, d_eff = [v for v in t if v != 0], [v for v in d if v != 0]
    else:
        t_eff, d_eff = t, d
    return edit_distance(d_eff, t_eff) / float(max_length)


def token_error_rate(y_true, decoded, tok_fn: Callable[[str], list], idx_to_text: Callable) -> float:
    """utils/metrics.py:42-59."""
    return error_rate(tok_fn(idx_to_text(y_true)), tok_fn(idx_to_text(decoded)))


def build_accuracy_fn(decode_fn):
    """utils/metrics.py:62-77: Accuracy(inputs, y_true) on the first utterance."""
    def Accuracy(inputs, y_true) -> float:  # the name is the key of the results dict (run_rnnt.py:323-324, :366)
        first = _as_list(y_true[0])
        decoded = decode_fn(inputs, max_length=len(first))
        return 1.0 - error_rate(first, decoded)
    return Accuracy


def build_wer_fn(decode_fn, idx_to_text: Callable):
    """utils/metrics.py:80-92: WER(inputs, y_true) on the first utterance, tokens = space-separated words."""
    def WER(inputs, y_true) -> float:
        first = _as_list(y_true[0])
        decoded = _as_list(decode_fn(inputs, max_length=len(first)))
        return token_error_rate(first, decoded, tok_fn=lambda t: t.split(" "), idx_to_text=idx_to_text)
    return WER
