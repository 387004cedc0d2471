"""The invariant the three-product form of the two-plane GEMMs rests on (csrc/gemm_planes.h).

csrc/engine_init.h::fill_planes stores a u16 weight q (model.cpp:610-616: w = q * scale + offset) as two fp16 planes
P_hi = fp16(q - 32896), P_lo = (q - 32896) - P_hi, and the kernels form a1 P_hi + a1 P_lo + a2 P_hi but not a2 P_lo.
That is only as accurate as four products if, for EVERY q,
  * P_hi + P_lo is exactly q - 32896 (the affine map's constants then need no change),
  * P_lo is itself an fp16 number (an integer of at most 16), and
  * |P_lo| <= 2^-11 |P_hi|, so that the product that is not formed is 2^-22 of the sum.
numpy's float16 is IEEE binary16 with round-to-nearest-even: the rounding f16_rne_bits (csrc/engine_init.h) implements."""
import numpy as np


def test_u16_weight_planes_are_exact_and_the_low_plane_is_eleven_bits_down():
    q = np.arange(65536, dtype=np.float64) - 32896.0
    hi = q.astype(np.float16)
    assert np.all(np.isfinite(hi))
    lo = q - hi.astype(np.float64)
    lo16 = lo.astype(np.float16)
    assert np.array_equal(lo16.astype(np.float64), lo)          # the remainder is an fp16 number ...
    assert np.array_equal(lo, np.round(lo)) and np.abs(lo).max() <= 16  # ... an integer of at most 16
    assert np.array_equal(hi.astype(np.float64) + lo16.astype(np.float64), q)  # the planes' sum is the file's integer
    nz = hi != 0
    assert np.all(np.abs(lo[nz]) <= np.abs(hi[nz].astype(np.float64)) * 2.0 ** -11)
    assert np.all(lo[~nz] == 0)
    # the planes of rounds 2-4 (the two bytes) would not do: their low plane is only eight bits down
    old_lo = (np.arange(65536) & 255) - 128.0
    old_hi = 256.0 * ((np.arange(65536) >> 8) - 128.0)
    assert np.array_equal(old_hi + old_lo, q)
    assert np.abs(old_lo).max() == 128 and np.abs(old_lo[old_hi != 0] / old_hi[old_hi != 0]).max() > 2.0 ** -9


def test_u8_weights_need_one_plane():
    q = np.arange(256, dtype=np.float64) - 128.0
    assert np.array_equal(q.astype(np.float16).astype(np.float64), q)
