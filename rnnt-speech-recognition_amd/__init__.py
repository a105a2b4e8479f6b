"""MI355X-native RNN-T joint + transducer-loss engine (drop-in for the loss path of
noahchalifour/rnnt-speech-recognition: utils/loss.py + the warp-transducer op behind it)."""
from .build import LIB_PATH, build  # noqa: F401
from .joint import JointLoss, joint_logits, rnnt_joint_loss  # noqa: F401
from .model import HParams, Transducer, TimeReduction, Encoder, PredictionNetwork  # noqa: F401
from .train import TrainStep, run_evaluate, run_training, synthetic_batch, synthetic_trained_like_joint  # noqa: F401
from .decoding import greedy_decode, greedy_decode_fn  # noqa: F401
from . import features, metrics, records  # noqa: F401
from .loss import RNNTLoss, get_loss_fn, reduced_lengths, rnnt_loss, rnnt_loss_and_grad  # noqa: F401

__all__ = ["rnnt_loss", "rnnt_loss_and_grad", "RNNTLoss", "get_loss_fn", "reduced_lengths", "rnnt_joint_loss",
           "joint_logits", "JointLoss", "build", "LIB_PATH"]
