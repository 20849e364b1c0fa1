"""Arithmetic identities the slice-resident kernel relies on (csrc/kernels_slice256.hip), checked in float32 on the CPU.

The kernel writes soft(a, c) as a - clamp(a, -c, c) (one v_med3_f32 and one subtraction) where the reference and the
other kernels use max(|a| - c, 0) * sign(a) (S1:18-19): the two must agree bit for bit, a zero's sign aside."""
import numpy as np


def soft_ref(a, c):
    m = np.abs(a) - c
    r = np.maximum(m, np.float32(0))
    return np.where(a < 0, -r, r).astype(np.float32)


def soft_med3(a, c):
    return (a - np.clip(a, -c, c)).astype(np.float32)


def test_soft_threshold_as_value_minus_clamp_is_bit_equal():
    rng = np.random.default_rng(5)
    for c in (np.float32(0.0015), np.float32(0.01125), np.float32(1.0 / 64), np.float32(0.3), np.float32(3.0)):
        a = np.concatenate([
            rng.standard_normal(200000).astype(np.float32) * c,                     # around the kink
            rng.standard_normal(200000).astype(np.float32),
            np.array([0.0, -0.0, c, -c, np.nextafter(c, np.float32(9)), np.nextafter(c, np.float32(0)),
                      -np.nextafter(c, np.float32(9)), 1e-38, -1e-38, 1e30, -1e30], np.float32)])
        r, m = soft_ref(a, c), soft_med3(a, c)
        assert np.array_equal(r, m)                                                 # -0.0 == +0.0 here, by design
        nz = r != 0
        assert np.array_equal(r[nz].view(np.uint32), m[nz].view(np.uint32))


def test_clip_equals_value_minus_soft():
    # S4:127-129: z - soft(z, 1/b) is the clamp the kernels compute directly
    rng = np.random.default_rng(6)
    ib = np.float32(1.0 / 64)
    z = (rng.standard_normal(300000) * 0.05).astype(np.float32)
    assert np.array_equal(np.clip(z, -ib, ib), (z - soft_ref(z, ib)).astype(np.float32))
