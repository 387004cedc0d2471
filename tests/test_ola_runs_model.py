"""The overlap-add schedule of the fused Wiener / inverse-STFT kernel (umx.cpp_amd/csrc/wiener_istft.h), restated in numpy
float32 and checked bit for bit against the reference's order (dsp.cpp:237-257: every output sample is the sum of the
frames that cover it, added in ascending frame order, then cropped by 2048 samples, dsp.cpp:203-205).

The device kernel gives every workgroup a RUN of consecutive frames.  A frame's four hop-sized chunks land on the hop blocks
f .. f + 3 of the padded signal; inside a run the three OPEN blocks (those later frames of the run still add to) are carried in
registers: chunk 3 is a block's first term (0 + c), chunks 2 and 1 are added to the carried sum, and when chunk 0 of the block's
last frame has been added the block is stored -- every stem sample ONCE (round 4; round 3 read-modify-wrote the stem per frame).
At the end of a run the open blocks are flushed.  The first three blocks of a run also belong to the previous run's last frames,
whose terms must come first: the run keeps its first three frames and `wiener_ola_edges_kernel` adds their chunks to those
blocks afterwards, in frame order, on top of what the previous run flushed.  This model runs the runs in an arbitrary order (workgroups are not ordered on the device) and must still give
the reference's bits; the GPU test of the kernel itself is tests/test_gpu_parity.py::test_fused_wiener_istft_equals_...
"""
import numpy as np
import pytest

HOP, NFFT = 1024, 4096


def reference_ola(frames, n):
    """dsp.cpp:237-257 + crop: out[s] = sum over f (ascending) of frames[f][s + 2048 - f * HOP]."""
    T = frames.shape[0]
    acc = np.zeros((T + 3) * HOP, np.float32)
    for f in range(T):  # ascending frame order = the order every sample's terms are added in
        acc[f * HOP:f * HOP + NFFT] = acc[f * HOP:f * HOP + NFFT] + frames[f]
    return acc[NFFT // 2:NFFT // 2 + n].copy()


def fused_ola(frames, n, run_len, rng):
    T = frames.shape[0]
    assert run_len >= 3  # csrc/engine_stages.h: a run that is not the last one has at least three frames
    stem = rng.standard_normal(n).astype(np.float32)  # whatever the buffer held before: must not matter
    kept = {}
    runs = [(f0, min(T, f0 + run_len)) for f0 in range(0, T, run_len)]

    stores = np.zeros(n, np.int32)  # how often the main kernel stores each stem sample

    def store(h, values):
        lo, hi = h * HOP - NFFT // 2, (h + 1) * HOP - NFFT // 2  # stem samples of hop block h
        a, b = max(lo, 0), min(hi, n)
        if a < b:
            stem[a:b] = values[a - lo:b - lo]
            stores[a:b] += 1

    for i in rng.permutation(len(runs)):  # workgroups run in no particular order
        f0, f1 = runs[i]
        open_ = np.zeros((3, HOP), np.float32)  # open_[c] = the sum so far of block f + 1 + c
        for f in range(f0, f1):  # ... but a workgroup takes its frames in order
            if f - f0 < 3:
                kept[f] = frames[f]
            nxt = np.zeros((3, HOP), np.float32)
            for c in range(4):
                a = open_[c] if c < 3 else np.zeros(HOP, np.float32)
                sm = a + frames[f][c * HOP:(c + 1) * HOP]
                if c == 0:
                    if f >= f0 + 3:  # block f is complete, and this run's alone
                        store(f, sm)
                else:
                    nxt[c - 1] = sm
            open_ = nxt
        for c in range(3):  # the run's end
            if f1 + c >= f0 + 3:
                store(f1 + c, open_[c])
    assert stores.max() <= 1  # each sample at most once by the main kernel
    for f0, _ in runs:  # wiener_ola_edges_kernel, after the main kernel
        for h in range(f0, f0 + 3):
            lo, hi = h * HOP - NFFT // 2, (h + 1) * HOP - NFFT // 2
            a, b = max(lo, 0), min(hi, n)
            if a >= b:
                continue
            fa, fb = max(0, h - 3), min(T - 1, h)
            acc = stem[a:b].copy() if fa < f0 else np.zeros(b - a, np.float32)
            for f in range(max(fa, f0), fb + 1):
                acc = acc + kept[f][(h - f) * HOP + a - lo:(h - f) * HOP + b - lo]
            stem[a:b] = acc
    return stem


@pytest.mark.parametrize("T,run_len", [(5, 5), (5, 3), (26, 3), (26, 4), (26, 7), (26, 81), (97, 13), (97, 96), (65, 9), (7, 3), (10, 9)])
def test_run_wise_overlap_add_has_the_bits_of_the_reference_order(T, run_len):
    rng = np.random.default_rng(1000 * T + run_len)
    # terms of very different magnitude, so that the order of the additions shows in the bits
    frames = (rng.standard_normal((T, NFFT)) * np.exp(rng.uniform(-12, 4, (T, NFFT)))).astype(np.float32)
    N = (T - 1) * HOP  # dsp.hpp:48: T = N / HOP + 1
    for n in (N, N - 777, max(1, N - 3 * HOP - 5), 1):
        ref = reference_ola(frames, n)
        got = fused_ola(frames, n, run_len, rng)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (T, run_len, n)


def test_a_different_order_of_the_terms_would_show():
    """The check above can fail: adding a block's terms in descending frame order changes bits."""
    rng = np.random.default_rng(7)
    T = 26
    frames = (rng.standard_normal((T, NFFT)) * np.exp(rng.uniform(-12, 4, (T, NFFT)))).astype(np.float32)
    n = (T - 1) * HOP
    acc = np.zeros((T + 3) * HOP, np.float32)
    for f in reversed(range(T)):
        acc[f * HOP:f * HOP + NFFT] = acc[f * HOP:f * HOP + NFFT] + frames[f]
    assert not np.array_equal(acc[NFFT // 2:NFFT // 2 + n].view(np.uint32), reference_ola(frames, n).view(np.uint32))


@pytest.mark.parametrize("T", [1, 4, 7, 9, 2584])
def test_window_sum_square_repeats_with_the_hop_on_interior_blocks(T):
    """wiener_istft_kernel takes the normalisation of an INTERIOR frame (hop blocks f .. f + 3 all complete: 3 <= f <= T - 4) from ONE
    hop of the window sum-square kept in LDS (block 3 of the table) instead of reading the table per sample: the table (dsp.hpp:80-101 as
    engine_init.h builds it -- frames in ascending order, w * w added in float32) must repeat with the hop, bit for bit, on every block
    that all four of its frames reach (3 <= h <= T - 1); the edge blocks differ and keep the per-sample read."""
    PI = np.float32(3.14159265359)
    n = np.arange(NFFT, dtype=np.float32)
    # dsp.hpp:61-78 in float32 (the device tables are built on the host with cosf; the bits of w do not matter for the property)
    w = (np.float32(0.5) * (np.float32(1.0) - np.cos(np.float32(2.0) * PI * n / np.float32(NFFT), dtype=np.float32))).astype(np.float32)
    w2 = w * w
    nw = np.zeros(NFFT + HOP * (T - 1), np.float32)
    for f in range(T):
        nw[f * HOP:f * HOP + NFFT] = nw[f * HOP:f * HOP + NFFT] + w2
    assert nw.size >= 4 * HOP  # the kernel reads block 3 whatever T is: in range
    blocks = nw.reshape(-1, HOP)
    interior = [h for h in range(blocks.shape[0]) if 3 <= h <= T - 1]
    for h in interior:
        assert np.array_equal(blocks[h].view(np.uint32), blocks[3].view(np.uint32)), h
    if T >= 4:
        assert not np.array_equal(blocks[2], blocks[3]) and not np.array_equal(blocks[T], blocks[3])  # the edges are different tables
    # the frames the kernel calls interior use only interior blocks
    for f in range(T):
        if f >= 3 and f + 3 <= T - 1:
            assert all(h in interior for h in range(f, f + 4))
