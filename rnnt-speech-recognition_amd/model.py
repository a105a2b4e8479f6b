"""The callers of the hot path: encoder, prediction network and TimeReduction as stock
PyTorch-ROCm modules (stock nn.LSTM with proj_size; PyTorch runs projected LSTMs through its own
per-timestep implementation, not MIOpen's fused one), wired to the fused joint + loss.

Reference: model.py:8-36 (TimeReduction), :39-81 (encoder), :84-116 (prediction network), :119-169
(build_keras_model); defaults hparams.py:3-37.  Nothing here is a custom kernel -- the point of this file is
that a user of the reference finds the same model surface in front of the MI355X loss engine.

Differences from the reference, all noted where they occur:
  * TimeReduction pads by (-T) mod f frames (the reference pads T mod f, model.py:33, which is only a valid
    reshape -- and then identical -- for f = 2, its default);
  * with `time_reduction_index` on the LAST encoder layer the reference's encoder output is f times wider than
    the prediction network's and the broadcast add at model.py:158-160 cannot work; this class raises;
  * the joint is the fused engine (joint.py) in `loss()`; `logits()` is the unfused form for decoding/tests;
  * layer defaults follow Keras / TF1, not torch: BatchNormalization eps 1e-3, momentum 0.99 (torch momentum 0.01);
    LayerNormalization eps 1e-3; tf.compat.v1 LSTMCell = glorot-uniform kernels, zero bias, forget_bias 1.0 (here folded
    into the forget-gate slice of bias_ih; torch's second bias vector is zero), bias-free projection;
  * torch's LSTM needs proj_size < hidden_size: for proj >= hidden (BASELINE configs 3/4: 320/320) the projection is
    dropped, whereas TF1's num_proj always projects -- same widths, one matrix fewer per layer;
  * `load_tf1_lstm_cell_` imports a TF1 LSTMCell's variables (gate order i, j, f, o; kernel [in + proj, 4 hidden]).
"""
from __future__ import annotations

import dataclasses
import json
import math
import os
from typing import Optional

import torch
from torch import nn

from .joint import JointLoss
from .loss import reduced_lengths


@dataclasses.dataclass
class HParams:
    """hparams.py:3-37 as plain data (no tensorboard)."""
    token_type: str = "word-piece"
    vocab_size: int = 4096
    mel_bins: int = 80
    frame_length: float = 0.025
    frame_step: float = 0.01
    hertz_low: float = 125.0
    hertz_high: float = 7600.0
    downsample_factor: int = 3
    embedding_size: int = 500
    encoder_layers: int = 8
    encoder_size: int = 2048
    projection_size: int = 640
    time_reduction_index: int = 1
    time_reduction_factor: int = 2
    pred_net_layers: int = 2
    pred_net_size: int = 2048
    joint_net_size: int = 640
    dropout: float = 0.0
    learning_rate: float = 1e-4

    def save(self, model_dir: str) -> None:  # utils/model.py:9-18 / run_rnnt.py:481
        with open(os.path.join(model_dir, "hparams.json"), "w") as f:
            json.dump(dataclasses.asdict(self), f, indent=1)

    @staticmethod
    def load(model_dir: str) -> "HParams":
        with open(os.path.join(model_dir, "hparams.json")) as f:
            return HParams(**json.load(f))


class TimeReduction(nn.Module):
    """Stack `factor` consecutive frames: [B, T, H] -> [B, ceil(T/factor), H*factor] (model.py:8-36)."""

    def __init__(self, factor: int):
        super().__init__()
        self.factor = int(factor)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, H = x.shape
        pad = (-T) % self.factor
        if pad:
            x = torch.nn.functional.pad(x, (0, 0, 0, pad))
        return x.reshape(B, (T + pad) // self.factor, H * self.factor)


class _LSTMBlock(nn.Module):
    """RNN(LSTMCell(d, num_proj)) -> Dropout -> LayerNorm  (model.py:62-68 / :104-109).  The TF1 LSTMCell
    projection has no bias, like torch's proj_size; padded frames are run through, as in the reference."""

    def __init__(self, in_size: int, hidden: int, proj: int, dropout: float):
        super().__init__()
        self.lstm = nn.LSTM(in_size, hidden, proj_size=proj if proj < hidden else 0, batch_first=True)
        self.drop = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(proj if proj < hidden else hidden, eps=1e-3)  # Keras LayerNormalization epsilon
        init_lstm_like_tf1_(self.lstm)

    def forward(self, x):
        y, _ = self.lstm(x)
        return self.norm(self.drop(y))


def _glorot_uniform_(w: torch.Tensor, fan_in: int, fan_out: int) -> None:
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        w.uniform_(-lim, lim)


def init_lstm_like_tf1_(lstm: nn.LSTM, forget_bias: float = 1.0) -> None:
    """tf.compat.v1.nn.rnn_cell.LSTMCell defaults (model.py:57-58, :101-102): ONE kernel [in + out, 4 hidden] drawn
    glorot-uniform, zero bias, forget_bias added to the forget gate at run time, bias-free projection kernel."""
    H = lstm.hidden_size
    out = lstm.proj_size or H
    for layer in range(lstm.num_layers):
        w_ih, w_hh = getattr(lstm, f"weight_ih_l{layer}"), getattr(lstm, f"weight_hh_l{layer}")
        fan_in = w_ih.shape[1] + out
        _glorot_uniform_(w_ih, fan_in, 4 * H)
        _glorot_uniform_(w_hh, fan_in, 4 * H)
        with torch.no_grad():
            getattr(lstm, f"bias_ih_l{layer}").zero_()
            getattr(lstm, f"bias_hh_l{layer}").zero_()
            getattr(lstm, f"bias_ih_l{layer}")[H:2 * H] = forget_bias  # torch gate order: i, f, g, o
        if lstm.proj_size:
            _glorot_uniform_(getattr(lstm, f"weight_hr_l{layer}"), H, lstm.proj_size)


def load_tf1_lstm_cell_(lstm: nn.LSTM, kernel, bias, projection_kernel=None, forget_bias: float = 1.0,
                        layer: int = 0) -> None:
    """Import the variables of one tf.compat.v1.nn.rnn_cell.LSTMCell (what the reference's checkpoints hold for every
    encoder / prediction-network layer, model.py:57-68, :101-109) into layer `layer` of a torch nn.LSTM.

      kernel            [in + out, 4 hidden]   applied to concat([x_t, m_{t-1}]); gate columns in TF1 order i, j, f, o
      bias              [4 hidden]             same order; forget_bias (1.0) is ADDED at run time, not stored
      projection_kernel [hidden, proj] or None (bias-free)
    torch: gates i, f, g, o (g = TF's j); weight_ih [4H, in], weight_hh [4H, out], two biases, weight_hr [proj, H]."""
    kernel = torch.as_tensor(kernel, dtype=torch.float32)
    bias = torch.as_tensor(bias, dtype=torch.float32)
    H = lstm.hidden_size
    w_ih, w_hh = getattr(lstm, f"weight_ih_l{layer}"), getattr(lstm, f"weight_hh_l{layer}")
    n_in, n_out = w_ih.shape[1], w_hh.shape[1]
    if tuple(kernel.shape) != (n_in + n_out, 4 * H) or tuple(bias.shape) != (4 * H,):
        raise ValueError(f"LSTMCell kernel/bias shapes {tuple(kernel.shape)}/{tuple(bias.shape)} do not fit "
                         f"[{n_in + n_out}, {4 * H}]/[{4 * H}]")
    if (projection_kernel is None) != (lstm.proj_size == 0):
        raise ValueError("projection kernel given for an LSTM without proj_size (or the reverse)")
    order = torch.cat([torch.arange(0, H), torch.arange(2 * H, 3 * H), torch.arange(H, 2 * H),
                       torch.arange(3 * H, 4 * H)])  # i, j, f, o -> i, f, j(=g), o
    kt = kernel[:, order].t().contiguous()             # [4H, in + out]
    b = bias[order].clone()
    b[H:2 * H] += forget_bias
    with torch.no_grad():
        w_ih.copy_(kt[:, :n_in])
        w_hh.copy_(kt[:, n_in:])
        getattr(lstm, f"bias_ih_l{layer}").copy_(b)
        getattr(lstm, f"bias_hh_l{layer}").zero_()
        if projection_kernel is not None:
            getattr(lstm, f"weight_hr_l{layer}").copy_(torch.as_tensor(projection_kernel, dtype=torch.float32).t())


class Encoder(nn.Module):
    def __init__(self, hp: HParams):
        super().__init__()
        feat = hp.mel_bins * hp.downsample_factor  # model.py:124
        # model.py:55: Keras BatchNormalization over the feature axis, epsilon 1e-3, momentum 0.99 (= torch momentum 0.01)
        self.input_norm = nn.BatchNorm1d(feat, eps=1e-3, momentum=0.01)
        self.blocks = nn.ModuleList()
        self.reduction_index = hp.time_reduction_index
        self.reduce = TimeReduction(hp.time_reduction_factor)
        width = feat
        for i in range(hp.encoder_layers):
            blk = _LSTMBlock(width, hp.encoder_size, hp.projection_size, hp.dropout)
            self.blocks.append(blk)
            width = blk.norm.normalized_shape[0]
            if i == hp.time_reduction_index:
                width *= hp.time_reduction_factor
        self.out_width = width

    def forward(self, mel_specs: torch.Tensor) -> torch.Tensor:
        x = self.input_norm(mel_specs.transpose(1, 2)).transpose(1, 2)
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i == self.reduction_index:
                x = self.reduce(x)
        return x


class PredictionNetwork(nn.Module):
    def __init__(self, hp: HParams):
        super().__init__()
        self.embed = nn.Embedding(hp.vocab_size, hp.embedding_size)  # model.py:99
        self.blocks = nn.ModuleList()
        width = hp.embedding_size
        for _ in range(hp.pred_net_layers):
            blk = _LSTMBlock(width, hp.pred_net_size, hp.projection_size, hp.dropout)
            self.blocks.append(blk)
            width = blk.norm.normalized_shape[0]
        self.out_width = width

    def forward(self, pred_inp: torch.Tensor) -> torch.Tensor:
        x = self.embed(pred_inp.long())  # the reference declares pred_inp float32 (model.py:132-133); ids either way
        for blk in self.blocks:
            x = blk(x)
        return x


class Transducer(nn.Module):
    """build_keras_model (model.py:119-169) with the joint fused into the loss."""

    def __init__(self, hp: HParams, blank_label: int = 0):
        super().__init__()
        self.hp = hp
        self.encoder = Encoder(hp)
        self.prediction = PredictionNetwork(hp)
        if self.encoder.out_width != self.prediction.out_width:
            raise ValueError(
                f"encoder output width {self.encoder.out_width} != prediction-network width "
                f"{self.prediction.out_width}: the broadcast add of model.py:158-160 needs equal widths (put "
                "time_reduction_index before the last encoder layer)")
        self.joint = JointLoss(self.encoder.out_width, hp.joint_net_size, hp.vocab_size, blank_label)

    def forward(self, mel_specs, pred_inp):
        """-> (enc [B, T', H], pred [B, U, H]); U = L_max + 1 because pred_inp = [0] ++ labels
        (utils/preprocessing.py:177-183)."""
        return self.encoder(mel_specs), self.prediction(pred_inp)

    def loss(self, mel_specs, pred_inp, spec_lengths, label_lengths, labels):
        """Per-utterance costs with the reference's argument set (run_rnnt.py:262-273)."""
        enc, pred = self(mel_specs, pred_inp)
        t_len = reduced_lengths(spec_lengths, self.hp.time_reduction_factor)  # utils/loss.py:31-33
        return self.joint(enc, pred, labels, t_len, label_lengths)

    def logits(self, mel_specs, pred_inp):
        enc, pred = self(mel_specs, pred_inp)
        return self.joint.logits(enc, pred)


def save_weights(model: nn.Module, path: str) -> None:
    """Weights-only checkpoint, like model.save_weights (run_rnnt.py:326-329): no optimizer state, step or RNG."""
    torch.save(model.state_dict(), path)


def load_weights(model: nn.Module, path: str, map_location: Optional[str] = None) -> None:
    model.load_state_dict(torch.load(path, map_location=map_location))
