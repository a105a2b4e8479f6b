"""Input side of the path: log-mel front end, text encoding and the 5-tensor training record.

Reference: utils/preprocessing.py:48-94 (compute_mel_spectrograms, downsample_spec), :177-183 (preprocess_text),
:236-296 (preprocess_audio / the record), utils/vocabulary.py:3-6 (character vocabulary), hparams.py:7-12 (front-end
defaults), run_rnnt.py:78-83 (zero-padded batches).  TensorFlow's signal ops are restated from their documented
definitions (tf.signal.stft: periodic Hann window, fft_length = next power of two, no end padding;
tf.signal.linear_to_mel_weight_matrix: HTK mel scale, triangles in the mel domain, DC bin excluded).
TensorFlow is not available in this image, so these are checked against the NumPy restatement in
oracle/features_oracle.py and against closed-form properties, not against TF outputs (parity unpinned, as for the loss).
`make_record` / `padded_batch` produce the five tensors from raw audio + text; the reference's `<name>.tfrecord` files of
those records are read and written by records.py (TFRecord framing, tf.train.Example and TensorProto wire formats,
utils/preprocessing.py:97-161), without TensorFlow.

Runs on whatever device the audio tensor lives on (cuFFT's ROCm counterpart through torch.fft on an MI355X)."""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch


def _next_pow2(n: int) -> int:
    return 1 << max(0, (int(n) - 1).bit_length())


def hertz_to_mel(f):
    """HTK mel scale used by tf.signal.linear_to_mel_weight_matrix."""
    return 1127.0 * torch.log1p(torch.as_tensor(f, dtype=torch.float64) / 700.0)


def linear_to_mel_weight_matrix(num_mel_bins: int, num_spectrogram_bins: int, sample_rate: float,
                                lower_edge_hertz: float, upper_edge_hertz: float) -> torch.Tensor:
    """[num_spectrogram_bins, num_mel_bins] triangular filterbank (f64 arithmetic, returned as f32)."""
    nyquist = sample_rate / 2.0
    lin = torch.linspace(0.0, nyquist, num_spectrogram_bins, dtype=torch.float64)[1:]  # DC bin left out
    spec_mel = hertz_to_mel(lin)[:, None]
    edges = torch.linspace(float(hertz_to_mel(lower_edge_hertz)), float(hertz_to_mel(upper_edge_hertz)),
                           num_mel_bins + 2, dtype=torch.float64)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = torch.clamp(torch.minimum(lower_slopes, upper_slopes), min=0.0)
    return torch.cat([torch.zeros(1, num_mel_bins, dtype=torch.float64), w], dim=0).to(torch.float32)


def stft_magnitude(audio: torch.Tensor, frame_length: int, frame_step: int) -> torch.Tensor:
    """|tf.signal.stft(audio, frame_length, frame_step)| : [frames, fft_length/2 + 1]; frames = 1 + (N - L) // step
    (no padding at the end), periodic Hann window, zero-padded to the next power of two."""
    audio = audio.to(torch.float32)
    n = audio.shape[-1]
    if n < frame_length:
        return audio.new_zeros((0, _next_pow2(frame_length) // 2 + 1))
    frames = audio.unfold(-1, frame_length, frame_step)  # [F, L]
    k = torch.arange(frame_length, device=audio.device, dtype=torch.float32)
    window = 0.5 - 0.5 * torch.cos(2.0 * math.pi * k / frame_length)  # periodic
    return torch.fft.rfft(frames * window, n=_next_pow2(frame_length), dim=-1).abs()


def compute_mel_spectrograms(audio_arr: torch.Tensor, sample_rate: int, n_mel_bins: int = 80,
                             frame_length: float = 0.025, frame_step: float = 0.01, hertz_low: float = 125.0,
                             hertz_high: float = 7600.0) -> torch.Tensor:
    """utils/preprocessing.py:48-81: log(mel + 1e-6) minus its per-bin mean over time (+1e-8)."""
    sr = float(sample_rate)
    fl, fs = int(round(sr * frame_length)), int(round(sr * frame_step))
    mag = stft_magnitude(audio_arr, fl, fs)
    mel_w = linear_to_mel_weight_matrix(n_mel_bins, mag.shape[-1], sr, hertz_low, hertz_high).to(mag.device)
    log_mel = torch.log(mag @ mel_w + 1e-6)
    return log_mel - (log_mel.mean(dim=0) + 1e-8)


def downsample_spec(mel_spec: torch.Tensor, n: int = 3) -> torch.Tensor:
    """utils/preprocessing.py:84-94: drop the tail that does not fill a group, stack n consecutive frames."""
    t, f = mel_spec.shape
    t3 = (t // n) * n
    return mel_spec[:t3].reshape(-1, f * n)


def preprocess_audio(audio: torch.Tensor, sample_rate: int, hp) -> torch.Tensor:
    """utils/preprocessing.py:236-253 with the front-end fields of model.HParams."""
    spec = compute_mel_spectrograms(audio, sample_rate, hp.mel_bins, hp.frame_length, hp.frame_step, hp.hertz_low,
                                    hp.hertz_high)
    return downsample_spec(spec, hp.downsample_factor)


# ---- text side ------------------------------------------------------------------------------------------------
def init_vocab() -> List[str]:
    """utils/vocabulary.py:3-8: index 0 is the blank ('')."""
    return ["", " ", "<s>", "</s>"] + list("abcdefghijklmnopqrstuvwxyz'")


def normalize_text(text: str) -> str:
    """utils/preprocessing.py:23-28."""
    return text.lower().replace('"', "")


class CharEncoder:
    """Character-level encoder over init_vocab() (utils/encoding.py:44-48 tf_vocab_encode = bytes_split + table lookup;
    the reference builds that table with default_value=0, utils/encoding.py:66-67: every byte outside the vocabulary
    -- digits, punctuation, each byte of a non-ASCII character -- becomes id 0, the blank)."""

    def __init__(self, vocab: Sequence[str] = None):
        self.vocab = list(vocab) if vocab is not None else init_vocab()
        self.index = {c: i for i, c in enumerate(self.vocab)}

    @property
    def vocab_size(self) -> int:
        return len(self.vocab)

    def encode(self, text: str) -> List[int]:
        # bytes_split: one token per UTF-8 byte (a non-ASCII character yields several unknown bytes -> several zeros)
        return [self.index.get(chr(b), 0) if b < 128 else 0 for b in text.encode("utf8")]

    def decode(self, ids) -> str:
        return "".join(self.vocab[int(i)] for i in ids if 0 <= int(i) < len(self.vocab))


def preprocess_text(text: str, encoder) -> Tuple[List[int], List[int]]:
    """utils/preprocessing.py:177-183: (labels, pred_inp = [0] ++ labels)."""
    enc = encoder.encode(normalize_text(text))
    return enc, [0] + enc


def make_record(audio: torch.Tensor, sample_rate: int, text: str, hp, encoder):
    """The 5 tensors of one training example (utils/preprocessing.py:283-289):
    (mel_specs f32 [T, mel_bins*downsample], pred_inp i32 [L+1], spec_length, label_length, labels i32 [L])."""
    mel = preprocess_audio(audio, sample_rate, hp)
    labels, pred_inp = preprocess_text(text, encoder)
    return (mel, torch.tensor(pred_inp, dtype=torch.int32), int(mel.shape[0]), len(labels),
            torch.tensor(labels, dtype=torch.int32))


def padded_batch(records):
    """dataset.padded_batch(batch_size, padded_shapes=([-1,-1],[-1],[],[],[-1])) (run_rnnt.py:78-83): zero padding to
    the longest example of the batch."""
    B = len(records)
    T = max(r[0].shape[0] for r in records)
    F = records[0][0].shape[1]
    U = max(r[1].shape[0] for r in records)
    L = max(max(r[4].shape[0] for r in records), 1)
    mel = torch.zeros(B, T, F)
    pred_inp = torch.zeros(B, U, dtype=torch.int32)
    labels = torch.zeros(B, L, dtype=torch.int32)
    for i, (m, pi, _, _, lab) in enumerate(records):
        mel[i, : m.shape[0]] = m
        pred_inp[i, : pi.shape[0]] = pi
        labels[i, : lab.shape[0]] = lab
    spec_lengths = torch.tensor([r[2] for r in records], dtype=torch.int32)
    label_lengths = torch.tensor([r[3] for r in records], dtype=torch.int32)
    return mel, pred_inp, spec_lengths, label_lengths, labels
