"""The hand-off layout of umx.cpp_amd/csrc/lstm_batch8.h restated in numpy: who publishes which granule, which thread polls it, where its
payload lands in LDS and which matrix-fragment element of which wave's B operand that is -- plus the arithmetic a gate lane sees.

The kernel's maps (one octet of 8 track lanes, one chain of Hl = 512 hidden units, 8 column shards of 64 units, 8 waves per workgroup):
  producer  lane (w, l) of shard s finishes unit U = 64 s + 8 w + 4 (n >> 3) + q of track n & 7 (n = l & 15, q = l >> 4); the lane with the
            even unit publishes the pair (U, U + 1) as granule g(U, track) = (U >> 3) * 32 + ((U & 7) >> 1) * 8 + track:
            {tag, h1(U) | h1(U+1) << 16, h2(U) | h2(U+1) << 16, 0}
  consumer  thread tid polls granules i * 512 + tid (i = 0 .. 3) and stores dword 1 / dword 2 at LDS byte
            ((ks * 4 + kq) * 16 + plane * 8 + track) * 16 + pair * 4 with ks = g >> 7, kq = (g >> 5) & 3, pair = (g >> 3) & 3, track = g & 7
  fragment  lane (n, q) of EVERY wave reads the 16 bytes at ((ks * 4 + q) * 16 + n) * 16 as the B fragment of k-step ks: element j is the
            fp16 of unit 32 ks + 8 q + j, plane n >> 3, track n & 7 (v_mfma_f32_16x16x32_f16: B[k = 8 q + j][column n])
This test builds the LDS image through the producer -> granule -> consumer path for random h and checks every fragment element against
that statement, that every granule has exactly one producer and one polling thread per workgroup, that a wave's 32 published granules are
512 contiguous bytes, and that the staged plane row (two waves x 64 lanes x 16 bytes) is the row the per-lane stores would write.
It also restates the per-cell arithmetic (columns n and n + 8 added, the affine map of model.cpp:610-616 on the sums) in float64 against
a direct dequantised dot product.  No GPU: the kernel itself is tested in tests/test_gpu_batch.py."""
import numpy as np

HL, UNITS, TRACKS = 512, 64, 8


def f16_planes(h):
    """h * 2^14 as two fp16 planes (csrc/gemm_planes.h split2_f16; the gate phase of the recurrence)."""
    x = (h.astype(np.float32) * np.float32(16384.0)).astype(np.float32)
    h1 = x.astype(np.float16)
    h2 = (x - h1.astype(np.float32)).astype(np.float16)
    return h1, h2


def producer_cell(shard, w, l):
    n, q = l & 15, l >> 4
    return 64 * shard + 8 * w + 4 * (n >> 3) + q, n & 7  # (unit, track)


def granule_of(unit, track):
    return (unit >> 3) * 32 + ((unit & 7) >> 1) * 8 + track


def test_every_granule_has_one_producer_one_poller_and_lands_in_fragment_order():
    rng = np.random.default_rng(5)
    h = rng.uniform(-1, 1, (HL, TRACKS)).astype(np.float32)  # h[unit][track]
    h1, h2 = f16_planes(h)
    ngran = HL // 2 * TRACKS
    gran = np.zeros((ngran, 4), np.uint32)
    writers = np.zeros(ngran, np.int32)
    for shard in range(HL // UNITS):
        for w in range(8):
            published = []
            for l in range(64):
                unit, track = producer_cell(shard, w, l)
                if unit & 1:
                    continue  # the odd unit's halves travel in its partner's granule (lane l ^ 16)
                pu, pt = producer_cell(shard, w, l ^ 16)
                assert (pu, pt) == (unit + 1, track)
                g = granule_of(unit, track)
                lo = lambda v: int(np.array(v, np.float16).view(np.uint16))
                gran[g] = (1, lo(h1[unit, track]) | lo(h1[unit + 1, track]) << 16, lo(h2[unit, track]) | lo(h2[unit + 1, track]) << 16, 0)
                writers[g] += 1
                published.append(g)
            assert sorted(published) == list(range(min(published), min(published) + 32))  # 512 contiguous bytes per wave
    assert (writers == 1).all()

    lds = np.zeros(16 * 1024, np.uint8)  # one step's h in fragment order
    polled = np.zeros(ngran, np.int32)
    for tid in range(512):
        for i in range(4):
            g = i * 512 + tid
            polled[g] += 1
            ks, kq, pair, track = g >> 7, (g >> 5) & 3, (g >> 3) & 3, g & 7
            assert (ks, kq, pair, track) == (i * 4 + (tid >> 7), (tid >> 5) & 3, (tid >> 3) & 3, tid & 7)  # the kernel's p_ks0 / p_q / p_pair / p_tr
            for plane in range(2):
                off = ((ks * 4 + kq) * 16 + plane * 8 + track) * 16 + pair * 4
                lds[off:off + 4] = np.array([gran[g][1 + plane]], np.uint32).view(np.uint8)
    assert (polled == 1).all()

    frag = lds.view(np.float16).reshape(HL // 32, 4, 16, 8)  # [ks][q][n][j]
    for ks in range(HL // 32):
        for q in range(4):
            for n in range(16):
                want = (h1 if n < 8 else h2)[32 * ks + 8 * q:32 * ks + 8 * q + 8, n & 7]
                assert (frag[ks, q, n].view(np.uint16) == want.view(np.uint16)).all(), (ks, q, n)


def test_staged_plane_row_is_the_row_of_the_per_lane_stores():
    rng = np.random.default_rng(6)
    shard = 3
    h = rng.uniform(-1, 1, (UNITS, TRACKS)).astype(np.float32)  # this workgroup's 64 units
    h1, h2 = f16_planes(h)
    stg = np.zeros((2, TRACKS, UNITS), np.uint16)  # [plane][track][unit of the shard]: the kernel's staging area
    for w in range(8):
        for l in range(64):
            unit, track = producer_cell(shard, w, l)
            ul = w * 8 + ((l & 15) >> 3) * 4 + (l >> 4)
            assert ul == unit - 64 * shard
            stg[0, track, ul] = h1[ul, track].view(np.uint16)
            stg[1, track, ul] = h2[ul, track].view(np.uint16)
    row = np.zeros((2, TRACKS, HL), np.uint16)  # what reaches the A planes: [plane][track lane][column dir * Hl + unit]
    for w in range(2):  # wave w stores plane w: lane l = 16 bytes = 8 units of one track
        for l in range(64):
            strk, sgrp = l >> 3, l & 7
            row[w, strk, shard * UNITS + sgrp * 8:shard * UNITS + sgrp * 8 + 8] = stg[w, strk, sgrp * 8:sgrp * 8 + 8]
    for track in range(TRACKS):
        assert (row[0, track, shard * UNITS:(shard + 1) * UNITS] == h1[:, track].view(np.uint16)).all()
        assert (row[1, track, shard * UNITS:(shard + 1) * UNITS] == h2[:, track].view(np.uint16)).all()


def test_cell_arithmetic_is_the_dequantised_dot_product():
    """(sum_k (q_k - 128) h1_k + sum_k (q_k - 128) h2_k) * wsc 2^-14 + (wof + 128 wsc) 2^-14 * sum_k h'_k = sum_k (q_k wsc + wof) h_k up to
    the 2^-22 split error of h (model.cpp:610-616 applied to the sum instead of per weight)."""
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, HL).astype(np.float64)
    wsc, wof = 0.0031, -0.41
    h = rng.uniform(-1, 1, HL).astype(np.float32)
    h1, h2 = f16_planes(h)
    s1 = float(np.sum((q - 128.0) * h1.astype(np.float64)))  # column n of the accumulator (exact products, fp32 accumulation on the device)
    s2 = float(np.sum((q - 128.0) * h2.astype(np.float64)))  # column n + 8
    hs = float(np.sum(h1.astype(np.float64) + h2.astype(np.float64)))
    got = (s1 + s2) * (wsc / 16384.0) + (wof + 128.0 * wsc) / 16384.0 * hs
    want = float(np.sum((q * wsc + wof) * h.astype(np.float64)))
    assert abs(got - want) < 1e-6 * max(1.0, abs(want)), (got, want)
    # the planes carry h to 22 significand bits
    assert np.abs((h1.astype(np.float64) + h2.astype(np.float64)) / 16384.0 - h.astype(np.float64)).max() < 2.0 ** -22
