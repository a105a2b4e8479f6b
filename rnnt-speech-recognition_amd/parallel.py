"""Utterance-sharded data parallelism for the loss path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What the reference does (its only parallelism):
  * tf.distribute.MirroredStrategy splits every GLOBAL batch contiguously over the replicas
    (run_rnnt.py:87-88 experimental_distribute_dataset),
  * each replica computes  loss = sum(costs_local) * (1 / GLOBAL_batch)      (run_rnnt.py:278),
  * optimizer.apply_gradients SUMS the replicas' gradients                  (run_rnnt.py:288),
  * the logged loss is the mean of the per-example costs over all replicas  (run_rnnt.py:293-294).
Utterances never communicate inside the loss, so the only exchange step is the gradient sum: ONE
all-reduce over a flat fp32 bucket (the joint's W1,b1,W2,b2 are ~1.7 MB at the reference defaults --
latency-bound on the xGMI mesh, so a single collective beats per-tensor calls).

The compute engine is injected (`costs_fn`), so the sharding/reduction logic is testable on CPU with
an oracle-backed engine; the product engines are rnnt_loss / rnnt_joint_loss (HIP only).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the global batch owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors: Sequence[torch.Tensor], world_size: int, rank: int) -> List[torch.Tensor]:
    """Slice every [global_B, ...] tensor to this rank's utterances."""
    gb = tensors[0].shape[0]
    lo, hi = shard_bounds(gb, world_size, rank)
    return [t[lo:hi] for t in tensors]


def balanced_order(input_lengths: torch.Tensor, label_lengths: torch.Tensor, world_size: int) -> torch.Tensor:
    """Permutation that deals utterances to ranks in descending lattice size (T_b * U_b), so ragged
    shards carry similar work (SURVEY.md 8e "sort/bucket by T_b*U_b").  Apply it to the global batch
    before shard_batch; costs come back in the permuted order."""
    work = input_lengths.to(torch.int64) * (label_lengths.to(torch.int64) + 1)
    order = torch.argsort(work, descending=True)
    gb = order.numel()
    slots: List[List[int]] = [[] for _ in range(world_size)]
    sizes = [shard_bounds(gb, world_size, r)[1] - shard_bounds(gb, world_size, r)[0] for r in range(world_size)]
    r, step = 0, 1
    for idx in order.tolist():  # snake deal, skipping full shards
        while len(slots[r]) >= sizes[r]:
            r += step
            if r == world_size or r < 0:
                step = -step
                r += step
        slots[r].append(idx)
        r += step
        if r == world_size or r < 0:
            step = -step
            r += step
    return torch.tensor([i for s in slots for i in s], dtype=torch.int64)


def flat_all_reduce_(tensors: Iterable[torch.Tensor], group=None) -> None:
    """SUM-all-reduce a list of tensors as ONE flat bucket (in place).  The bucket takes the tensors' dtype
    (fp32 for every product parameter; mixed lists are promoted to the widest)."""
    ts = [t for t in tensors if t is not None]
    if not ts or not dist.is_initialized():  # a one-rank group still runs the collective: the same code path at every N
        return
    dtype = ts[0].dtype
    for t in ts[1:]:
        dtype = torch.promote_types(dtype, t.dtype)
    flat = torch.cat([t.reshape(-1).to(dtype) for t in ts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in ts:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def dp_loss_step(costs_fn: Callable[[], torch.Tensor], params: Sequence[torch.Tensor], global_batch: int,
                 group=None) -> torch.Tensor:
    """One data-parallel loss+grad step with the reference's semantics.

    costs_fn() -> per-utterance costs of THIS rank's shard (differentiable w.r.t. `params`).
    After the call every p.grad holds d/dp [ sum over the GLOBAL batch of costs / global_batch ], identical
    on all ranks.  Returns the logged loss (mean cost over the global batch), identical on all ranks."""
    for p in params:
        p.grad = None
    costs = costs_fn()
    local = costs.sum() * (1.0 / global_batch)  # run_rnnt.py:278 -- GLOBAL batch in the denominator
    local.backward()
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat_all_reduce_([p.grad for p in params], group)  # run_rnnt.py:288 -- replicas' gradients are summed
    logged = local.detach().clone()
    if dist.is_initialized():
        dist.all_reduce(logged, op=dist.ReduceOp.SUM, group=group)  # run_rnnt.py:293-294
    return logged
