"""CPU tests of the rows either side of the hot path (SURVEY.md 8f-3 input format, 8f-4 decode + metrics)."""
import math

import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg
from rnnt_speech_recognition_amd import decoding, features, metrics
from oracle import features_oracle as fo


# ---------------------------------------------------------------- f-3: log-mel front end + record
def test_log_mel_matches_numpy_restatement():
    rng = np.random.default_rng(0)
    sr = 16000
    t = np.arange(int(0.73 * sr)) / sr
    audio = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * rng.normal(size=t.size)).astype(np.float32)
    got = features.compute_mel_spectrograms(torch.tensor(audio), sr).numpy()
    ref = fo.log_mel(audio, sr)
    assert got.shape == ref.shape == (1 + (audio.size - 400) // 160, 80)
    assert np.abs(got - ref).max() < 2e-3  # f32 FFT + log of small mel energies
    assert np.abs(got.mean(axis=0)).max() < 1e-4  # per-bin mean removed


def test_mel_matrix_properties():
    W = features.linear_to_mel_weight_matrix(80, 257, 16000.0, 125.0, 7600.0).numpy()
    assert W.shape == (257, 80) and not W[0].any() and (W >= 0).all()
    np.testing.assert_allclose(W, fo.mel_matrix(80, 257, 16000.0, 125.0, 7600.0), atol=1e-6)
    centers = W.argmax(axis=0)
    assert (np.diff(centers) >= 0).all() and centers[-1] > centers[0]  # triangles march up the spectrum


def test_downsample_and_record():
    hp = pkg.HParams()
    audio = torch.randn(16000)
    enc = features.CharEncoder()
    assert enc.vocab_size == 31 and enc.vocab[0] == ""  # blank at index 0 (utils/vocabulary.py:3-6)
    mel, pred_inp, spec_len, label_len, labels = features.make_record(audio, 16000, 'He said "hi"', hp, enc)
    frames = 1 + (16000 - 400) // 160
    assert mel.shape == (frames // 3, 240) and spec_len == frames // 3
    assert labels.tolist() == enc.encode("he said hi") and pred_inp.tolist() == [0] + labels.tolist()
    assert label_len == len("he said hi") and enc.decode(labels) == "he said hi"
    np.testing.assert_allclose(features.downsample_spec(torch.arange(14.0).reshape(7, 2)).numpy(),
                               fo.downsample(np.arange(14.0).reshape(7, 2)))
    short = features.make_record(torch.randn(9000), 16000, "ok", hp, enc)
    mel_b, pi_b, sl, ll, lab_b = features.padded_batch([(mel, pred_inp, spec_len, label_len, labels), short])
    assert mel_b.shape == (2, spec_len, 240) and pi_b.shape == (2, 11) and lab_b.shape == (2, 10)
    assert not mel_b[1, short[2]:].any() and sl.tolist() == [spec_len, short[2]] and ll.tolist() == [10, 2]


# ---------------------------------------------------------------- f-4: metrics
def test_edit_distance_known_values():
    assert metrics.edit_distance("kitten", "sitting") == 3
    assert metrics.edit_distance([], [1, 2]) == 2 and metrics.edit_distance([1, 2, 3], [1, 2, 3]) == 0
    assert metrics.edit_distance([1, 2, 3], [2, 3]) == 1


def test_error_rate_semantics():
    # zeros (padding / blank) are dropped from id sequences but count in the dense length (utils/metrics.py:8-23)
    assert metrics.error_rate([[5, 6, 7, 0, 0]], [[5, 7]]) == pytest.approx(1 / 5)
    assert metrics.error_rate([[1, 2]], [[1, 2]]) == 0.0
    assert metrics.error_rate(["a", "b", "c"], ["a", "c"]) == pytest.approx(1 / 3)
    enc = features.CharEncoder()
    ids = lambda s: enc.encode(s)
    wer = metrics.token_error_rate(ids("the cat sat"), ids("the bat sat down"), lambda t: t.split(" "), enc.decode)
    assert wer == pytest.approx(2 / 4)


# ---------------------------------------------------------------- f-4: greedy decode
def small_model(seed=0):
    torch.manual_seed(seed)
    hp = pkg.HParams(vocab_size=12, mel_bins=4, downsample_factor=2, embedding_size=8, encoder_layers=2,
                     encoder_size=16, projection_size=8, time_reduction_index=0, pred_net_layers=1, pred_net_size=16,
                     joint_net_size=64)
    return pkg.Transducer(hp)


def brute_force_decode(model, mel, max_length):
    """The reference's formulation, spelled out: full prediction-network re-run per joint evaluation."""
    model.eval()
    with torch.no_grad():
        enc = model.encoder(mel[:1])
        hyp = [0]
        for i in range(enc.shape[1]):
            while True:
                g = model.prediction(torch.tensor([hyp]))[:, -1:, :]
                k = int(model.joint.logits(enc[:, i : i + 1], g)[0, 0, 0].argmax())
                if k == 0:
                    break
                hyp.append(k)
                if max_length is not None and len(hyp) >= max_length + 1:
                    return hyp[1:]
        return hyp[1:]


@pytest.mark.parametrize("max_length", [None, 3, 40])
def test_greedy_decode_matches_stateless_reference_form(max_length):
    model = small_model(3)
    with torch.no_grad():
        model.joint.b2[0] -= 0.4  # make blanks a little rarer so that symbols are emitted
    mel = torch.randn(2, 30, 8)
    cap = max_length if max_length is not None else 60  # an untrained model may never emit blank: bound the test
    a = decoding.greedy_decode(model, mel, cap).tolist()[0]
    b = decoding.greedy_decode(model, mel, cap, stateless=True).tolist()[0]
    c = brute_force_decode(model, mel, cap)
    model.train()
    assert decoding.greedy_decode(model, mel, cap).tolist()[0] == a  # train-mode caller: decode still runs in eval mode
    assert a == b == c
    assert all(k != 0 for k in a) and len(a) <= cap
    assert model.training  # mode restored


def test_accuracy_and_wer_builders():
    model = small_model(5)
    mel = torch.randn(1, 20, 8)
    y_true = torch.tensor([[3, 4, 5, 0, 0]])
    dec = decoding.greedy_decode_fn(model)
    acc = metrics.build_accuracy_fn(dec)(mel, y_true)
    hyp = dec(mel, max_length=5).tolist()[0]
    assert acc == pytest.approx(1.0 - metrics.error_rate([3, 4, 5, 0, 0], hyp))
    vocab = ["", " "] + list("abcdefghij")
    to_text = lambda ids: "".join(vocab[int(i)] for i in ids)
    w = metrics.build_wer_fn(dec, to_text)(mel, y_true)
    assert 0.0 <= w <= max(1.0, float(len(hyp)))


def test_char_encoder_maps_unknown_bytes_to_zero_like_the_reference():
    """utils/encoding.py:66-67: build_lookup_table(vocab, default_value=0); tf_vocab_encode splits BYTES (:44-48)."""
    enc = pkg.features.CharEncoder()
    ids = enc.encode("a1b!é c")
    assert min(ids) >= 0
    a, b, c, sp = (enc.index[ch] for ch in "abc ")
    assert ids == [a, 0, b, 0, 0, 0, sp, c]  # digit, punctuation -> 0; 'é' is two UTF-8 bytes -> two zeros
    hp = pkg.HParams(vocab_size=enc.vocab_size, mel_bins=8, downsample_factor=3)
    rec = pkg.features.make_record(torch.randn(4000), 16000, "route 66, ok!", hp, enc)
    assert int(rec[1].min()) >= 0 and int(rec[4].min()) >= 0  # pred_inp feeds an Embedding: never negative
