"""Train-step harness with the reference's step semantics (run_rnnt.py:253-298, :483-488):

  costs = loss_fn(labels, model(mel_specs, pred_inp), spec_lengths, label_lengths)      :269-273
  loss  = sum(costs) * (1 / GLOBAL_batch)                                              :278
  gradients summed across replicas, SGD(lr = 1e-4, momentum = 0.9)                      :288, :483-484
  logged loss = mean of the per-example costs over all replicas                        :293-294

One process per GPU; utterance-sharded data parallelism through parallel.dp_loss_step (ONE RCCL all-reduce
of the flat gradient bucket).  The fp16 / LossScaleOptimizer branch of the reference (:275-276, :486-488) is not
reproduced: its native op is float32-only and nothing casts the fp16 logits, so that path cannot have worked.
"""
from __future__ import annotations

import time
from typing import Dict, Sequence

import torch
import torch.distributed as dist

from . import parallel
from .model import Transducer


class TrainStep:
    def __init__(self, model: Transducer, global_batch: int, learning_rate: float = None, momentum: float = 0.9,
                 group=None):
        self.model = model
        self.global_batch = int(global_batch)
        self.group = group
        self.params = [p for p in model.parameters() if p.requires_grad]
        lr = model.hp.learning_rate if learning_rate is None else learning_rate
        self.optimizer = torch.optim.SGD(self.params, lr=lr, momentum=momentum)
        self.step_count = 0

    def __call__(self, mel_specs, pred_inp, spec_lengths, label_lengths, labels) -> Dict[str, float]:
        """`inputs` are THIS rank's shard of the global batch (parallel.shard_batch).  Returns the log fields of
        run_rnnt.py:360-364 (loss, step time)."""
        t0 = time.time()
        self.model.train()
        logged = parallel.dp_loss_step(
            lambda: self.model.loss(mel_specs, pred_inp, spec_lengths, label_lengths, labels),
            self.params, self.global_batch, self.group)
        self.optimizer.step()
        self.step_count += 1
        return {"loss": float(logged), "step_time": time.time() - t0, "step": self.step_count}

    @torch.no_grad()
    def evaluate(self, mel_specs, pred_inp, spec_lengths, label_lengths, labels) -> float:
        """Eval loss of run_rnnt.py:392-424 (mean cost over the global batch); metrics/decoding are out of scope."""
        self.model.eval()
        costs = self.model.loss(mel_specs, pred_inp, spec_lengths, label_lengths, labels)
        s = costs.sum() / self.global_batch
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(s, group=self.group)
        return float(s)


def synthetic_batch(hp, batch: int, frames: int, max_labels: int, device, seed: int = 1234, ragged: bool = True):
    """A batch with the reference's five tensors (utils/preprocessing.py:110-161) from random log-mel-like
    features: (mel_specs f32 [B, T, mel*down], pred_inp i32 [B, L+1], spec_lengths, label_lengths, labels)."""
    g = torch.Generator().manual_seed(seed)
    feat = hp.mel_bins * hp.downsample_factor
    mel = torch.randn(batch, frames, feat, generator=g)
    labels = torch.randint(1, hp.vocab_size, (batch, max_labels), generator=g, dtype=torch.int32)
    if ragged:
        spec_len = torch.randint(frames // 2, frames + 1, (batch,), generator=g, dtype=torch.int32)
        lab_len = torch.randint(max_labels // 2, max_labels + 1, (batch,), generator=g, dtype=torch.int32)
        spec_len[0], lab_len[0] = frames, max_labels
    else:
        spec_len = torch.full((batch,), frames, dtype=torch.int32)
        lab_len = torch.full((batch,), max_labels, dtype=torch.int32)
    for b in range(batch):  # padded_batch pads with zeros (run_rnnt.py:78-83)
        mel[b, spec_len[b]:] = 0.0
        labels[b, lab_len[b]:] = 0
    pred_inp = torch.cat([torch.zeros(batch, 1, dtype=torch.int32), labels], dim=1)  # [0] ++ labels
    return [t.to(device) for t in (mel, pred_inp, spec_len, lab_len, labels)]
