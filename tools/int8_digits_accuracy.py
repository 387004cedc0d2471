"""tools/int8_digits_accuracy.py -- CPU only: what would the int8-digit arithmetic planned for the dense stack cost
or gain in accuracy against today's two fp16 planes?

Both schemes multiply the file's integer weights exactly and apply the affine map q*s+o to the accumulated sum, so what differs
is (a) how an activation row is represented and (b) how the products are accumulated:
  two fp16 planes (csrc/gemm_planes.h)   row scaled by a power of two into [2^14, 2^15); a1 = fp16(a'), a2 = fp16(a' - a1);
                                         products exact, fp32 accumulation on the matrix cores
  three int8 digits (planned)            row scaled into [2^22, 2^23), rounded to an integer, cut into balanced digits
                                         d2*2^16 + d1*2^8 + d0, |d| <= 128; products AND sums exact in int32
Emulated in numpy on a UMX-L-shaped fc1 (K = 2974, u8 weights) with spectrogram-like inputs; reference = float64 with the
weights the reference uses, fl32(q*s+o) (model.cpp:610-616).  The fp32 accumulation of the matrix cores is emulated as a
running fp32 sum over k in blocks of 16 (the instruction's K), which is what bounds the planes' result from below.
"""
import numpy as np

import sys

rng = np.random.default_rng(5)
M, K, N = 256, 2974, 512
if len(sys.argv) > 1 and sys.argv[1] == "bounded":
    # activations as W_ih / fc2 see them: tanh / LSTM outputs, |a| < 1, no heavy tail
    x = np.tanh(rng.normal(0.0, 0.8, (M, K))).astype(np.float32)
    print("# activations: tanh-like, |a| < 1 (W_ih, fc2)")
else:
    # magnitudes with the dynamic range of a spectrogram row after the input scaling (F8): log-normal, a few large bins
    x = (np.exp(rng.normal(-3.0, 2.0, (M, K))) * rng.choice([1.0, -1.0], (M, K))).astype(np.float32)
    print("# activations: spectrogram-like, log-normal magnitudes (fc1)")
q = rng.integers(0, 256, (N, K)).astype(np.int64)
s, o = np.float32(0.0123 / 255), np.float32(-0.0061)
w_ref = (q.astype(np.float32) * s + o).astype(np.float32)  # what the reference multiplies
want = x.astype(np.float64) @ w_ref.astype(np.float64).T


def affine(int_sum, rowsum, unscale):
    # sum_k a_k (q_k s + o) = s * sum_k a_k (q_k - 128) + (o + 128 s) * sum_k a_k
    return (np.float32(s) * unscale[:, None]).astype(np.float32) * int_sum.astype(np.float32) + (np.float32(o + 128 * s) * rowsum)[:, None]


rowsum = x.sum(axis=1, dtype=np.float32)
mx = np.abs(x).max(axis=1)

# ---- two fp16 planes
e = 15 - np.ceil(np.log2(mx)).astype(int)  # max into [2^14, 2^15)
xs = x * np.exp2(e)[:, None].astype(np.float32)
a1 = xs.astype(np.float16)
a2 = (xs - a1.astype(np.float32)).astype(np.float16)
acc = np.zeros((M, N), np.float32)
wq = (q - 128).astype(np.float32)
for k0 in range(0, K, 16):  # fp32 accumulation, one matrix instruction's K at a time, smaller term first
    blk = slice(k0, min(K, k0 + 16))
    acc = acc + (a2[:, blk].astype(np.float64) @ wq[:, blk].astype(np.float64).T).astype(np.float32)
    acc = acc + (a1[:, blk].astype(np.float64) @ wq[:, blk].astype(np.float64).T).astype(np.float32)
planes = affine(acc, rowsum, np.exp2(-e).astype(np.float32))

# ---- three balanced int8 digits, exact integer accumulation
e8 = 23 - np.ceil(np.log2(mx)).astype(int)  # max into [2^22, 2^23)
xi = np.rint(x.astype(np.float64) * np.exp2(e8)[:, None]).astype(np.int64)
d0 = ((xi + 128) & 255) - 128
r1 = (xi - d0) >> 8
d1 = ((r1 + 128) & 255) - 128
d2 = (r1 - d1) >> 8
assert np.abs(d2).max() <= 128 and np.array_equal(d2 * 65536 + d1 * 256 + d0, xi)
wi = q - 128
sums = [d @ wi.T for d in (d0, d1, d2)]
assert max(np.abs(t).max() for t in sums) < 2 ** 31  # int32 accumulators are enough
comb = (sums[2].astype(np.float32) * np.float32(65536) + sums[1].astype(np.float32) * np.float32(256)) + sums[0].astype(np.float32)
digits = affine(comb, rowsum, np.exp2(-e8).astype(np.float32))

# ---- plain fp32 (the reference's own arithmetic, CPU order)
f32 = np.zeros((M, N), np.float32)
for k0 in range(0, K, 16):
    blk = slice(k0, min(K, k0 + 16))
    f32 = f32 + (x[:, blk].astype(np.float64) @ w_ref[:, blk].astype(np.float64).T).astype(np.float32)

for name, got in (("two fp16 planes (today)", planes), ("three int8 digits (planned)", digits), ("fp32 products, fp32 sum", f32)):
    err = got.astype(np.float64) - want
    print(f"{name:30s} rel L2 {np.linalg.norm(err) / np.linalg.norm(want):9.3e}   max abs {np.abs(err).max():9.3e}")
