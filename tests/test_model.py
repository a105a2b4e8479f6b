"""The callers of the hot path: model surface on CPU, train step on GPU."""
import math

import numpy as np
import pytest
import torch

import rnnt_speech_recognition_amd as pkg


def small_hp(**kw):
    d = dict(vocab_size=28, mel_bins=8, downsample_factor=3, embedding_size=16, encoder_layers=2, encoder_size=48,
             projection_size=32, time_reduction_index=0, time_reduction_factor=2, pred_net_layers=1, pred_net_size=48,
             joint_net_size=64, learning_rate=1e-3)
    d.update(kw)
    return pkg.HParams(**d)


def test_time_reduction_matches_reference_reshape():
    # model.py:8-36 with factor 2: pad T mod 2 frames, then stack pairs of frames along the feature axis
    x = torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 5, 3)
    y = pkg.TimeReduction(2)(x)
    assert y.shape == (2, 3, 6)
    assert torch.equal(y[0, 0], torch.cat([x[0, 0], x[0, 1]]))
    assert torch.equal(y[0, 2], torch.cat([x[0, 4], torch.zeros(3)]))
    assert pkg.TimeReduction(2)(torch.zeros(1, 4, 3)).shape == (1, 2, 6)
    # lengths seen by the loss follow the same ceil (utils/loss.py:31-33)
    assert pkg.reduced_lengths(torch.tensor([5, 4]), 2).tolist() == [3, 2]


def test_model_shapes_and_width_check():
    hp = small_hp()
    m = pkg.Transducer(hp)
    mel = torch.randn(3, 11, 24)
    pred_inp = torch.randint(0, 28, (3, 6))
    enc, pred = m(mel, pred_inp)
    assert enc.shape == (3, 6, 32) and pred.shape == (3, 6, 32)  # T' = ceil(11/2); U = L+1 = 6
    assert m.logits(mel, pred_inp).shape == (3, 6, 6, 28)
    with pytest.raises(ValueError, match="broadcast add"):
        pkg.Transducer(small_hp(time_reduction_index=1))  # reduction after the LAST layer: reference bug, refused


def test_hparams_defaults_are_the_reference_ones(tmp_path):
    hp = pkg.HParams()
    assert (hp.mel_bins, hp.downsample_factor, hp.embedding_size, hp.encoder_layers, hp.encoder_size,
            hp.projection_size, hp.time_reduction_index, hp.time_reduction_factor, hp.pred_net_layers,
            hp.pred_net_size, hp.joint_net_size, hp.learning_rate, hp.vocab_size) == \
           (80, 3, 500, 8, 2048, 640, 1, 2, 2, 2048, 640, 1e-4, 4096)  # hparams.py:3-37
    hp.save(str(tmp_path))
    assert pkg.HParams.load(str(tmp_path)) == hp


@pytest.mark.parametrize("proj", [None, 5])
def test_tf1_lstm_cell_weights_import(proj):
    """model.py:57-68: a reference checkpoint holds tf.compat.v1 LSTMCell variables (gate order i, j, f, o, forget_bias
    added at run time, bias-free projection); imported into torch's nn.LSTM the layer must compute the same sequence."""
    from oracle import lstm_oracle
    from rnnt_speech_recognition_amd.model import load_tf1_lstm_cell_

    rng = np.random.default_rng(5)
    n_in, H = 7, 9
    out = proj or H
    kernel = rng.normal(size=(n_in + out, 4 * H)) * 0.4
    bias = rng.normal(size=4 * H) * 0.3
    pk = rng.normal(size=(H, proj)) * 0.5 if proj else None
    x = rng.normal(size=(3, 11, n_in))
    ref = lstm_oracle.tf1_lstm_cell_sequence(x, kernel, bias, pk)
    lstm = torch.nn.LSTM(n_in, H, proj_size=proj or 0, batch_first=True).double()
    load_tf1_lstm_cell_(lstm, kernel, bias, pk)
    y, _ = lstm(torch.tensor(x))
    np.testing.assert_allclose(y.detach().numpy(), ref, atol=1e-6, rtol=0)  # the importer stores float32 weights
    with pytest.raises(ValueError):
        load_tf1_lstm_cell_(lstm, kernel[:, :-4], bias, pk)


def test_layer_defaults_follow_keras_and_tf1():
    m = pkg.Transducer(small_hp())
    bn = m.encoder.input_norm
    assert (bn.eps, bn.momentum) == (1e-3, 0.01)  # Keras BatchNormalization: epsilon 1e-3, momentum 0.99
    blk = m.encoder.blocks[0]
    assert blk.norm.eps == 1e-3                    # Keras LayerNormalization epsilon
    H = blk.lstm.hidden_size
    b = blk.lstm.bias_ih_l0.detach()
    assert torch.all(b[H:2 * H] == 1.0) and torch.all(b[:H] == 0) and torch.all(b[2 * H:] == 0)  # forget_bias = 1.0
    assert torch.all(blk.lstm.bias_hh_l0 == 0)
    lim = (6.0 / (blk.lstm.weight_ih_l0.shape[1] + blk.lstm.proj_size + 4 * H)) ** 0.5  # glorot over the ONE TF kernel
    assert blk.lstm.weight_ih_l0.abs().max() <= lim and blk.lstm.weight_hh_l0.abs().max() <= lim


def test_evaluate_runs_the_metric_builders(monkeypatch):
    """run_rnnt.py:392-424: eval step = loss + run_metrics(mel_specs, labels).  The loss engine is HIP-only, so it is
    stubbed here; what is checked is the wiring of build_accuracy_fn / build_wer_fn into TrainStep.evaluate."""
    hp = small_hp()
    m = pkg.Transducer(hp)
    batch = pkg.synthetic_batch(hp, batch=3, frames=12, max_labels=4, device="cpu", seed=2)
    monkeypatch.setattr(pkg.Transducer, "loss", lambda self, *a: torch.tensor([3.0, 6.0, 9.0]))
    step = pkg.TrainStep(m, global_batch=3)
    dec = pkg.greedy_decode_fn(m)
    enc = pkg.features.CharEncoder()
    acc = pkg.metrics.build_accuracy_fn(dec)
    wer = pkg.metrics.build_wer_fn(dec, lambda ids: enc.decode(ids))
    loss, res = step.evaluate(*batch, metrics=[acc, wer])
    assert loss == pytest.approx(6.0)
    assert set(res) == {"Accuracy", "WER"} and all(0.0 <= v <= 1.0 or v >= 0 for v in res.values())
    # the same numbers as calling the metric on its own (first utterance of the shard, run_rnnt.py:223-230)
    assert res["Accuracy"] == pytest.approx(acc(batch[0], batch[4]))


def test_evaluate_is_the_mean_over_the_examples_present(monkeypatch):
    """strategy.reduce(MEAN, loss, axis=0) (run_rnnt.py:417-418): a short final evaluation batch is divided by ITS size,
    not by the configured global batch (records.batches(drop_remainder=False) yields one)."""
    hp = small_hp()
    m = pkg.Transducer(hp)
    step = pkg.TrainStep(m, global_batch=8)
    batch = pkg.synthetic_batch(hp, batch=3, frames=12, max_labels=4, device="cpu", seed=2)
    monkeypatch.setattr(pkg.Transducer, "loss", lambda self, *a: torch.tensor([3.0, 6.0, 9.0]))
    assert step.evaluate(*batch)[0] == pytest.approx(6.0)
    full = [batch, batch]
    from rnnt_speech_recognition_amd import train as tr
    assert tr.run_evaluate(step, iter(full))[0] == pytest.approx(6.0)


@pytest.mark.gpu
def test_train_step_reduces_loss_and_matches_unfused_loss():
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    hp = small_hp()
    m = pkg.Transducer(hp).to(dev)
    batch = pkg.synthetic_batch(hp, batch=6, frames=40, max_labels=7, device=dev, seed=3)
    mel, pred_inp, spec_len, lab_len, labels = batch
    # fused loss == reference composition (materialised logits -> get_loss_fn adapter)
    m.eval()
    fused = m.loss(*batch)
    unfused = pkg.get_loss_fn(hp.time_reduction_factor)(labels, m.logits(mel, pred_inp), spec_len, lab_len)
    np.testing.assert_allclose(fused.detach().cpu().numpy(), unfused.detach().cpu().numpy(), rtol=1e-4)
    step = pkg.TrainStep(m, global_batch=6)
    first = step(*batch)["loss"]
    losses = [first]
    for _ in range(40):
        losses.append(step(*batch)["loss"])
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * first, losses[::8]
    assert abs(step.evaluate(*batch)[0] - losses[-1]) < 0.5 * first


@pytest.mark.gpu
def test_word_piece_sized_model_trains_through_the_f16_joint():
    """Vocabulary 512 / joint width 128: JointLoss picks the f16-MFMA joint on its own (reference: --fp16_run with the
    word-piece vocabulary, hparams.py:5, run_rnnt.py:96-99).  Fused costs stay within binary16 distance of the unfused
    f32 composition, SGD makes progress, greedy decoding runs on the trained model."""
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    hp = pkg.HParams(vocab_size=512, mel_bins=4, downsample_factor=2, embedding_size=16, encoder_layers=2,
                     encoder_size=32, projection_size=16, time_reduction_index=0, pred_net_layers=1, pred_net_size=32,
                     joint_net_size=128, learning_rate=1e-3)
    m = pkg.Transducer(hp).to(dev)
    batch = pkg.synthetic_batch(hp, batch=4, frames=36, max_labels=6, device=dev, seed=5)
    mel, pred_inp, spec_len, lab_len, labels = batch
    m.eval()
    fused = m.loss(*batch)
    unfused = pkg.get_loss_fn(hp.time_reduction_factor)(labels, m.logits(mel, pred_inp), spec_len, lab_len)
    np.testing.assert_allclose(fused.detach().cpu().numpy(), unfused.detach().cpu().numpy(), rtol=1e-4)
    step = pkg.TrainStep(m, global_batch=4)
    losses = [step(*batch)["loss"] for _ in range(30)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses[::6]
    hyp = pkg.greedy_decode(m, mel, max_length=8)
    assert hyp.shape[0] == 1 and hyp.shape[1] <= 8 and int(hyp.min() if hyp.numel() else 1) > 0


@pytest.mark.gpu
def test_c3_end_to_end_step_at_baseline_size():
    """BASELINE.json configs[2] at its own size: B=64, 600 frames x 240 features, encoder 2 x 320 (x2 time reduction
    after layer 0), prediction network 1 x 320, joint 320, V=28.  The fused HIP loss and its gradients w.r.t. the joint
    weights are checked against the float64 oracle on two utterances (fed with the model's own encoder / prediction
    outputs), then three SGD steps run at full size."""
    from oracle import rnnt_oracle as orc

    torch.manual_seed(1234)
    dev = torch.device("cuda:0")
    hp = pkg.HParams(vocab_size=28, embedding_size=320, encoder_layers=2, encoder_size=320, projection_size=320,
                     time_reduction_index=0, pred_net_layers=1, pred_net_size=320, joint_net_size=320)
    m = pkg.Transducer(hp).to(dev)
    batch = pkg.synthetic_batch(hp, batch=64, frames=600, max_labels=100, device=dev, seed=77)
    mel, pred_inp, spec_len, lab_len, labels = batch
    m.train()  # (the RNN backward needs training mode; dropout is 0, BatchNorm uses the batch statistics in both forwards)
    picks = [0, 17]
    mask = torch.zeros(64, device=dev)
    mask[picks] = 1.0
    costs = m.loss(*batch)
    (costs * mask).sum().backward()
    enc, pred = m(mel, pred_inp)
    assert enc.shape == (64, 300, 320) and pred.shape == (64, 101, 320)
    t_len = pkg.reduced_lengths(spec_len, 2).cpu().numpy()
    j = m.joint
    n = lambda x: x.detach().cpu().numpy()
    ref = orc.joint_loss_and_grads(n(enc)[picks], n(pred)[picks], n(j.W1), n(j.b1), n(j.W2), n(j.b2), n(labels)[picks],
                                   t_len[picks], n(lab_len)[picks])
    np.testing.assert_allclose(n(costs)[picks], ref["costs"], rtol=1e-4)
    for got, key in ((j.W1.grad, "dW1"), (j.b1.grad, "db1"), (j.W2.grad, "dW2"), (j.b2.grad, "db2")):
        assert np.abs(n(got) - ref[key]).max() <= 1e-4 * max(1.0, np.abs(ref[key]).max()), key
    m.zero_grad()
    step = pkg.TrainStep(m, global_batch=64)  # the reference's SGD(1e-4, momentum 0.9)
    losses = [step(*batch)["loss"] for _ in range(4)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    ev, _ = step.evaluate(*batch)
    assert np.isfinite(ev)


def test_training_loop_follows_the_reference_schedule(tmp_path):
    """run_rnnt.py:300-377 with a stand-in step (the loop itself needs no GPU): evaluation + checkpoint before every
    steps_per_checkpoint-th step (also before step 0) and once after the last epoch; log lines; epoch means."""
    from rnnt_speech_recognition_amd import train as tr

    class FakeStep:
        group = None
        model = torch.nn.Linear(1, 1)  # run_evaluate averages its buffers once per evaluation (no group here: no-op)

        def __init__(self):
            self.n, self.saved, self.evals = 0, [], 0

        def __call__(self, *inputs):
            self.n += 1
            return {"loss": float(inputs[0]), "step_time": 0.001, "step": self.n}

        def evaluate(self, *inputs, metrics=None, sync_buffers=True):
            assert sync_buffers is False  # run_evaluate synchronised once, up front
            self.evals += 1
            return float(inputs[0]) * 2, {fn.__name__: fn(None, None) for fn in (metrics or [])}

        def save_checkpoint(self, path):
            self.saved.append(path)

    def Accuracy(_x, _y):
        return 0.25

    step, lines = FakeStep(), []
    batches = lambda: iter([(1.0, 0, 0, 0, 0), (2.0, 0, 0, 0, 0), (6.0, 0, 0, 0, 0)])
    evals = lambda: iter([(5.0, 0, 0, 0, 0), (7.0, 0, 0, 0, 0)])
    out = tr.run_training(step, batches, n_epochs=2, steps_per_log=2, steps_per_checkpoint=4, eval_batches=evals,
                          eval_metrics=[Accuracy], checkpoint_template=str(tmp_path / "ck_{step}_{val_loss:.1f}.pt"),
                          log=lines.append)
    assert step.n == 6 and out["steps"] == 6 and out["loss"] == pytest.approx(3.0)
    # checkpoints before global steps 0 and 4 and after the last epoch: 3 evaluations of 2 batches each
    assert step.evals == 6 and [p.split("ck_")[1] for p in step.saved] == ["0_12.0.pt", "4_12.0.pt", "6_12.0.pt"]
    assert out["val_loss"] == pytest.approx(12.0) and out["val_Accuracy"] == pytest.approx(0.25)
    assert lines[0] == "Starting training." and lines[1].startswith("VALIDATION RESULTS: Time: ")
    assert "Loss: 12.0000, Accuracy: 0.2500" in lines[1] and lines[2].startswith("Saving checkpoint ")
    logs = [l for l in lines if l.startswith("Epoch: ")]
    assert [l.split(", Step Time")[0] for l in logs] == ["Epoch: 0, Batch: 0, Global Step: 0", "Epoch: 0, Batch: 2, Global Step: 2",
                                                        "Epoch: 1, Batch: 1, Global Step: 4"]
    assert logs[1].endswith("Loss: 3.0000")  # running mean of 1, 2, 6 within the epoch
    assert [l for l in lines if l.startswith("EPOCH RESULTS")] == ["EPOCH RESULTS: Loss: 3.0000"] * 2
    loss, res = tr.run_evaluate(step, evals(), [Accuracy])
    assert loss == pytest.approx(12.0) and res == {"Accuracy": 0.25}


@pytest.mark.gpu
def test_training_loop_from_tfrecords_on_gpu(tmp_path):
    """The callers either side of the path, end to end: reference-format TFRecord files -> padded batches -> train steps
    through the fused HIP joint + loss -> evaluation with the decode metrics -> weights-only checkpoint."""
    import rnnt_speech_recognition_amd as pkg
    from rnnt_speech_recognition_amd import decoding, features, metrics, records

    torch.manual_seed(0)
    enc = features.CharEncoder()
    hp = pkg.HParams(vocab_size=enc.vocab_size, mel_bins=8, downsample_factor=3, embedding_size=16, encoder_layers=2,
                     encoder_size=32, projection_size=16, time_reduction_index=0, pred_net_layers=1, pred_net_size=32,
                     joint_net_size=64)
    texts = ["hello world", "ok", "the cat", "a b c d", "yes", "no way"]
    recs = [features.make_record(torch.randn(6000 + 900 * i), 16000, t, hp, enc) for i, t in enumerate(texts)]
    records.write_dataset(recs[:4], str(tmp_path / "train.tfrecord"))
    records.write_dataset(recs[4:], str(tmp_path / "dev.tfrecord"))
    dev = torch.device("cuda:0")
    model = pkg.Transducer(hp).to(dev)
    step = pkg.TrainStep(model, global_batch=2)
    to_dev = lambda t5: tuple(x.to(dev) if torch.is_tensor(x) else x for x in t5)
    dec = decoding.greedy_decode_fn(model)
    lines = []
    out = pkg.run_training(
        step, lambda: records.batches(records.load_dataset(str(tmp_path), "train"), 2), n_epochs=2, steps_per_log=1,
        steps_per_checkpoint=2, eval_batches=lambda: records.batches(records.load_dataset(str(tmp_path), "dev"), 2),
        eval_metrics=[metrics.build_accuracy_fn(dec)], checkpoint_template=str(tmp_path / "model_{step}_{val_loss:.2f}.pt"),
        to_device=to_dev, log=lines.append)
    assert out["steps"] == 4 and math.isfinite(out["loss"]) and math.isfinite(out["val_loss"]) and "val_Accuracy" in out
    assert len(list(tmp_path.glob("model_*.pt"))) >= 2 and sum(l.startswith("Epoch: ") for l in lines) == 4
