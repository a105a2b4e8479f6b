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
from typing import Callable, Dict, Iterable, Optional, Sequence, Tuple

import math

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
        # MirroredStrategy keeps every variable mirrored from creation (run_rnnt.py:119-122, :455-462): replicas must
        # start from rank 0's weights and buffers, whatever seed each process happened to use.
        sync_replicas_(model, group)
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
    def evaluate(self, mel_specs, pred_inp, spec_lengths, label_lengths, labels,
                 metrics: Optional[Iterable[Callable]] = None, sync_buffers: bool = True) -> Tuple[float, Dict[str, float]]:
        """One eval step of run_rnnt.py:392-424: (mean cost over the examples of this global batch, {metric name: value}).

        `metrics` are the callables of metrics.build_accuracy_fn / build_wer_fn: each is called as
        metric_fn(mel_specs, labels) (run_metrics, run_rnnt.py:223-230) on this rank's shard -- they decode its first
        utterance -- and reduced with MEAN across replicas (:421-422).  BatchNorm running statistics are averaged across
        replicas first so that every rank evaluates the same model (`sync_buffers=False`: the caller has done that once for
        the whole evaluation, as run_evaluate does)."""
        if sync_buffers:
            sync_buffers_(self.model, self.group)
        self.model.eval()
        costs = self.model.loss(mel_specs, pred_inp, spec_lengths, label_lengths, labels)
        # strategy.reduce(MEAN, loss, axis=0) (run_rnnt.py:417-418): the mean over the examples actually present on all
        # replicas -- a short final batch (records.batches(drop_remainder=False)) is divided by its own size
        sc = torch.stack([costs.sum().to(torch.float64), torch.tensor(float(costs.numel()), dtype=torch.float64,
                                                                      device=costs.device)])
        multi = dist.is_initialized() and dist.get_world_size(self.group) > 1
        if multi:
            dist.all_reduce(sc, group=self.group)
        s = sc[0] / sc[1].clamp_min(1.0)
        results: Dict[str, float] = {}
        for fn in (metrics or []):
            v = torch.tensor(float(fn(mel_specs, labels)), dtype=torch.float64, device=costs.device)
            if multi:
                dist.all_reduce(v, group=self.group)
                v = v / dist.get_world_size(self.group)
            results[fn.__name__] = float(v)
        return float(s), results

    def save_checkpoint(self, path: str) -> None:
        """Weights-only checkpoint (run_rnnt.py:326-329), written by rank 0 after the replicas' buffers are averaged."""
        from .model import save_weights

        sync_buffers_(self.model, self.group)
        if not dist.is_initialized() or dist.get_rank(self.group) == 0:
            save_weights(self.model, path)


class _Mean:
    """tf.keras.metrics.Mean: running average of the values it is called with."""
    def __init__(self):
        self.total, self.count = 0.0, 0

    def __call__(self, value):
        self.total += float(value)
        self.count += 1

    def result(self) -> float:
        return self.total / self.count if self.count else 0.0


def run_evaluate(step: "TrainStep", eval_batches: Iterable, metrics: Optional[Iterable[Callable]] = None,
                 to_device=None) -> Tuple[float, Dict[str, float]]:
    """run_rnnt.py:380-452: the mean over the evaluation batches of (loss, every metric).  `eval_batches` yields this
    rank's 5-tuples (features.padded_batch / records.batches); `to_device` moves one to the model's device."""
    metrics = list(metrics or [])
    loss_object = _Mean()
    metric_objects = {fn.__name__: _Mean() for fn in metrics}
    sync_buffers_(step.model, step.group)  # once per evaluation: nothing trains in between
    for inputs in eval_batches:
        if to_device is not None:
            inputs = to_device(inputs)
        loss, results = step.evaluate(*inputs, metrics=metrics, sync_buffers=False)
        loss_object(loss)
        for name, value in results.items():
            metric_objects[name](value)
    return loss_object.result(), {name: obj.result() for name, obj in metric_objects.items()}


def run_training(step: "TrainStep", train_batches: Callable[[], Iterable], n_epochs: int, steps_per_log: int = 1,
                 steps_per_checkpoint: int = 1000, eval_batches: Optional[Callable[[], Iterable]] = None,
                 eval_metrics: Optional[Iterable[Callable]] = None, checkpoint_template: Optional[str] = None,
                 to_device=None, log: Callable[[str], None] = print) -> Dict[str, float]:
    """The reference's training loop around the step (run_rnnt.py:300-377), line for line in behaviour:

      * `train_batches()` / `eval_batches()` return a fresh iterator per epoch / per evaluation (a tf.data dataset is
        re-iterable; a generator is not);
      * evaluation + checkpoint BEFORE every step whose global index is a multiple of `steps_per_checkpoint` -- hence also
        before the first step -- when an evaluation set is given (:347-349), and once more after the last epoch (:377);
      * every `steps_per_log` steps one line with the epoch's running mean loss and the step time (:360-365), one
        'EPOCH RESULTS' line per epoch (:372-375), one 'VALIDATION RESULTS' line per evaluation (:314-318);
      * checkpoints are weights only, named `checkpoint_template.format(step=..., val_loss=...)` (:326-329).
    TensorBoard summaries (:320-325, :367-370) are not written.  Returns the last epoch's mean loss and the last
    validation results."""
    is_rank0 = not dist.is_initialized() or dist.get_rank(step.group) == 0
    say = log if is_rank0 else (lambda _msg: None)
    last: Dict[str, float] = {}

    def checkpoint_model(global_step: int) -> None:
        t0 = time.time()
        eval_loss, results = run_evaluate(step, eval_batches(), eval_metrics, to_device)
        line = "VALIDATION RESULTS: Time: {:.4f}, Loss: {:.4f}".format(time.time() - t0, eval_loss)
        for name, value in results.items():
            line += ", {}: {:.4f}".format(name, value)
        say(line)
        last.update({"val_loss": eval_loss, **{"val_" + k: v for k, v in results.items()}})
        if checkpoint_template is not None:
            path = checkpoint_template.format(step=global_step, val_loss=eval_loss)
            say("Saving checkpoint {}".format(path))
            step.save_checkpoint(path)

    say("Starting training.")
    global_step = 0
    for epoch in range(n_epochs):
        loss_object = _Mean()
        for batch, inputs in enumerate(train_batches()):
            if global_step % steps_per_checkpoint == 0 and eval_batches is not None:
                checkpoint_model(global_step)
            if to_device is not None:
                inputs = to_device(inputs)
            out = step(*inputs)
            loss_object(out["loss"])
            if global_step % steps_per_log == 0:
                say("Epoch: {}, Batch: {}, Global Step: {}, Step Time: {:.4f}, Loss: {:.4f}".format(
                    epoch, batch, global_step, out["step_time"], loss_object.result()))
            global_step += 1
        say("EPOCH RESULTS: Loss: {:.4f}".format(loss_object.result()))
        last["loss"] = loss_object.result()
    if eval_batches is not None:
        checkpoint_model(global_step)
    last["steps"] = float(global_step)
    return last


def _flat_collective_(tensors: Sequence[torch.Tensor], fn) -> None:
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    if not ts:
        return
    for dtype in sorted({t.dtype for t in ts}, key=str):  # the same bucket order on every rank
        same = [t for t in ts if t.dtype == dtype]
        flat = torch.cat([t.detach().reshape(-1) for t in same])
        fn(flat)
        off = 0
        for t in same:
            n = t.numel()
            t.detach().copy_(flat[off:off + n].view_as(t))
            off += n


def sync_replicas_(model: torch.nn.Module, group=None, src: int = 0) -> None:
    """Broadcast every parameter and buffer from the group's rank `src` (one flat bucket per dtype).  No-op without a group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    # dist.broadcast takes a GLOBAL rank: translate the group-local source (a sub-group need not contain global rank 0)
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    _flat_collective_(list(model.parameters()) + list(model.buffers()),
                      lambda flat: dist.broadcast(flat, src=gsrc, group=group))


def sync_buffers_(model: torch.nn.Module, group=None) -> None:
    """Average the floating-point buffers (BatchNorm running mean / variance) across replicas; integer buffers
    (num_batches_tracked) are identical by construction.  No-op without a group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    n = dist.get_world_size(group)

    def mean_(flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(n)

    _flat_collective_([b for b in model.buffers() if b.is_floating_point()], mean_)


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


def synthetic_trained_like_joint(B: int, T: int, U: int, V: int, J: int, seed: int = 7, gain: float = 10.0, late_every: int = 2,
                                 input_lengths=None, label_lengths=None):
    """Projections and output-layer weights of a joint network whose posteriors look like a TRAINED transducer's: one dominant
    symbol per lattice cell along a monotone alignment -- blank until the cell's label is due, the label afterwards (bonus `gain`
    nats, as tests/test_peaky_gpu.py builds them on logits) -- under N(0,1)-sized noise from the remaining joint units.  Every
    `late_every`-th utterance emits all its labels in the last 40 % of its frames (alignments far from the lattice's diagonal).
    What the headline's N(0,1) inputs do not show: the alignment band is narrow, so the backward's row pruning skips most rows.

    Construction (exact in the reference's joint, model.py:158-166): M = min(V - 1, J // 2) "symbol units"; unit k = v mod M of
    symbol v carries tanh(a (t - emit_u)) where the cell's label belongs to it (enc_proj = a (t - T/2), pred_proj = -a (emit_u - T/2))
    and -1 elsewhere (pred_proj = -40), W2[k, v] = gain; the other J - M units are N(0,1) projections with glorot weights.
    input_lengths / label_lengths (optional, [B]): the utterances' frames T_b / labels L_b -- emission times fall inside T_b.
    Returns CPU tensors (enc_proj [B,T,J], pred_proj [B,U,J], W2 [J,V], b2 [V], labels i32 [B,U-1])."""
    g = torch.Generator().manual_seed(seed)
    M = min(V - 1, J // 2)
    a = min(0.1, 76.0 / T)  # |a (t - T/2)| stays inside the tanh tables' range (|x| <= 43, csrc/rnnt_common.h)
    labels = torch.randint(1, V, (B, U - 1), generator=g, dtype=torch.int32)
    ep = torch.randn(B, T, J, generator=g)
    pp = torch.randn(B, U, J, generator=g)
    lim = math.sqrt(6.0 / (J + V))
    W2 = (torch.rand(J, V, generator=g) * 2 - 1) * lim
    b2 = torch.zeros(V)
    W2[:M] = 0.0
    for v in range(1, V):
        W2[v % M, v] = gain
    t = torch.arange(T, dtype=torch.float32)
    ep[:, :, :M] = (a * (t - T / 2.0))[None, :, None]
    pp[:, :, :M] = -40.0
    for b in range(B):
        Tb = int(input_lengths[b]) if input_lengths is not None else T
        Lb = int(label_lengths[b]) if label_lengths is not None else U - 1
        lo = int(0.6 * Tb) if (late_every and b % late_every == late_every - 1) else 0
        emit = torch.sort(torch.randint(lo, max(Tb, lo + 1), (Lb,), generator=g)).values.to(torch.float32)
        k = (labels[b, :Lb].to(torch.int64) % M)
        pp[b, torch.arange(Lb), k] = -a * (emit - T / 2.0)
    return ep, pp, W2, b2, labels
