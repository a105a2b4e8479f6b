"""CPU check of the finding behind the float64 recurrence of the log-domain sweeps (csrc/rnnt_sweep.h alpha_sweep_pr; DESIGN.md 4
Numerics): on a wide lattice with fewer frames than columns under 4 x N(0,1) logits, a float32 recurrence -- whatever the
granularity of its integer offsets -- ends 1e-4 ... 7e-4 from the float64 oracle, and the same recurrence in float64 over the SAME
float32 edge weights, with float32-grade stores, ends below 2e-5.  NumPy emulation (tests/tools/emulate_sweep.py), no GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import emulate_sweep as em  # noqa: E402


def test_float32_recurrence_is_the_error_source_on_wide_peaked_lattices():
    c_ref, out = em.run("sigma4", 40, 941, 7, seed=3, rules=("lane16", "lane1", "lane16f64"))
    f32_lane16, f32_lane1, f64_lane16 = (out[r][1] for r in ("lane16", "lane1", "lane16f64"))
    assert f32_lane16 > 1e-4 and f32_lane1 > 1e-4, out   # finer offsets do not help ...
    assert f64_lane16 < 2e-5, out                        # ... a float64 recurrence on the same inputs does
    assert all(out[r][0] < 1e-6 for r in out), out       # costs are fine either way


def test_f16_joint_shapes_are_padded_to_the_native_grid():
    """Host logic of rnnt_joint_loss (joint.py padded_joint_shape): V to the next multiple of 128 (at least 128), J to the next
    multiple of 128 up to 640 (round 4; before: V to multiples of 512, J to 128 / 256 / 512 / 640)."""
    import pytest
    from rnnt_speech_recognition_amd.joint import padded_joint_shape

    assert padded_joint_shape(640, 4096, "f16") == (640, 4096)
    assert padded_joint_shape(320, 1000, "f16") == (384, 1024)
    assert padded_joint_shape(100, 33, "f16") == (128, 128)
    assert padded_joint_shape(384, 640, "f16") == (384, 640)
    assert padded_joint_shape(320, 28, "f32") == (320, 28)
    assert padded_joint_shape(100, 31, "f32") == (128, 31)
    with pytest.raises(ValueError):
        padded_joint_shape(700, 512, "f16")
    with pytest.raises(ValueError):
        padded_joint_shape(128, 9000, "f16")
    # round 5: the f32-grade joint takes up to 128 symbols (vocabulary tiles of 32) at joint sizes up to 640, 32 beyond
    assert padded_joint_shape(128, 40, "f32") == (128, 40)
    assert padded_joint_shape(600, 128, "f32") == (640, 128)
    with pytest.raises(ValueError):
        padded_joint_shape(128, 129, "f32")
    with pytest.raises(ValueError):
        padded_joint_shape(704, 40, "f32")
    from rnnt_speech_recognition_amd.joint import _auto_joint_dtype
    assert [_auto_joint_dtype(640, v) for v in (28, 32, 33, 64, 65, 128, 4096)] == ["f32", "f32", "f32", "f32", "f16", "f16", "f16"]
    assert _auto_joint_dtype(704, 40) == "f16"
