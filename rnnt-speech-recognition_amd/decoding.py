"""Greedy transducer decoding (reference utils/decoding.py:6-108).

Behaviour kept from the reference:
  * only the FIRST utterance of the batch is decoded (utils/decoding.py:22,35);
  * the hypothesis starts with the blank/start token 0 (:28) and the returned ids drop it (:102);
  * per encoder frame, symbols are emitted until the joint's argmax is blank (id 0) (:61-76); there is no
    per-frame symbol cap, only the global `max_length` (:78-79): once max_length symbols exist, decoding stops;
  * the encoder runs in inference mode (BatchNorm statistics frozen, no dropout) (:37,63).
The reference re-runs the prediction network over the whole hypothesis for every joint evaluation (:63-64); it is a
causal LSTM stack, so carrying its state forward gives the same outputs at O(1) per symbol.  `stateless=True` keeps
the reference's formulation (used by the tests as the cross-check).

The joint is evaluated for ONE lattice cell at a time (T = U = 1), as the reference does (`joint`, utils/decoding.py:6-18).
On the device it goes through the engine's logits-only entry point (compute_rnnt_joint_logits via JointLoss.cell_logits):
the forward kernels of the fused loss, so the decoder sees the logits the loss was trained on."""
from __future__ import annotations

from typing import List, Optional

import torch


@torch.no_grad()
def greedy_decode(model, mel_specs: torch.Tensor, max_length: Optional[int] = None, stateless: bool = False) -> torch.Tensor:
    """mel_specs [B, T, F] -> int32 ids [1, n] of the first utterance (blank-free, start token removed)."""
    was_training = model.training
    model.eval()
    try:
        x = mel_specs[:1]
        enc = model.encoder(x)  # [1, T', H]
        dev = enc.device
        hyp: List[int] = [0]
        pred_net = model.prediction
        joint = model.joint

        # incremental prediction-network state: one (h, c) per LSTM block
        states = [None] * len(pred_net.blocks)

        def pred_last_incremental(token: int) -> torch.Tensor:
            y = pred_net.embed(torch.tensor([[token]], device=dev))
            for i, blk in enumerate(pred_net.blocks):
                y, states[i] = blk.lstm(y, states[i])
                y = blk.norm(blk.drop(y))
            return y  # [1, 1, H]

        def pred_last_stateless() -> torch.Tensor:
            return pred_net(torch.tensor([hyp], device=dev))[:, -1:, :]

        g = pred_last_stateless() if stateless else pred_last_incremental(0)
        max_reached = False
        for i in range(enc.shape[1]):
            if max_reached:
                break
            f = enc[:, i : i + 1, :]
            while True:
                # (consumed by the argmax below before the next call: the engine may hand out its cached buffer)
                logits = joint.cell_logits(f, g, reuse_buffers=True)[0, 0, 0]  # [V]
                k = int(torch.argmax(torch.log_softmax(logits, dim=-1)).item())
                if k == joint.blank_label:
                    break
                hyp.append(k)
                g = pred_last_stateless() if stateless else pred_last_incremental(k)
                if max_length is not None and len(hyp) >= max_length + 1:
                    max_reached = True
                    break
        return torch.tensor([hyp[1:]], dtype=torch.int32, device=dev)
    finally:
        model.train(was_training)


def greedy_decode_fn(model):
    """The reference's factory shape: greedy_decode_fn(model, hparams) -> fn(inputs, max_length)."""
    def fn(inputs: torch.Tensor, max_length: Optional[int] = None) -> torch.Tensor:
        return greedy_decode(model, inputs, max_length)
    return fn
