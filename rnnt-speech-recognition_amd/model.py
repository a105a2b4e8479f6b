"""The callers of the hot path: encoder, prediction network and TimeReduction as stock
PyTorch-ROCm modules (MIOpen LSTM with projection), wired to the fused joint + loss.

Reference: model.py:8-36 (TimeReduction), :39-81 (encoder), :84-116 (prediction network), :119-169
(build_keras_model); defaults hparams.py:3-37.  Nothing here is a custom kernel -- the point of this file is
that a user of the reference finds the same model surface in front of the MI355X loss engine.

Differences from the reference, all noted where they occur:
  * TimeReduction pads by (-T) mod f frames (the reference pads T mod f, model.py:33, which is only a valid
    reshape -- and then identical -- for f = 2, its default);
  * with `time_reduction_index` on the LAST encoder layer the reference's encoder output is f times wider than
    the prediction network's and the broadcast add at model.py:158-160 cannot work; this class raises;
  * the joint is the fused engine (joint.py) in `loss()`; `logits()` is the unfused form for decoding/tests.
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
        self.norm = nn.LayerNorm(proj if proj < hidden else hidden)

    def forward(self, x):
        y, _ = self.lstm(x)
        return self.norm(self.drop(y))


class Encoder(nn.Module):
    def __init__(self, hp: HParams):
        super().__init__()
        feat = hp.mel_bins * hp.downsample_factor  # model.py:124
        self.input_norm = nn.BatchNorm1d(feat)     # model.py:55 (Keras BatchNormalization over the feature axis)
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
