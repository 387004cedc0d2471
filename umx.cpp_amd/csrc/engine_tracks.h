// engine_tracks.h -- part of engine.hip (one translation unit: the kernels inline into their launchers): whole tracks on the device: split_inference / shift_inference for one track or one per lane.
// Included by engine.hip behind the definition of umx_hip_ctx; not a stand-alone header.
// ---------------------------------------------------------------- whole track
// shift_inference (umx.cpp:99-150) around split_inference (umx.cpp:152-295) with the track resident in HBM:
// one upload, the segments queued back to back through the two pipeline slots (so consecutive segments
// overlap exactly as in bench.py), the weighted overlap-add and the final normalisation on the device, one
// download.  shift_offset < 0: no shift buffer (plain split_inference).
int umx_hip_ctx::track(const float *audio_host, int length, int shift_offset, float *const out_host[4], unsigned flags,
                       void (*progress)(float, void *), void *progress_user)
{
    if (!out_host)
    {
        set_error("track: need audio, outputs, length >= 1 and shift offset < 22050");
        return UMX_ERR_ARG;
    }
    return tracks(1, &audio_host, &length, &shift_offset, out_host, flags, progress, progress_user);
}

// shift_inference (umx.cpp:99-150) around split_inference (umx.cpp:152-295) for `nt` tracks at once, one per track
// lane: the tracks stay in HBM, call s runs segment s of every track that still has one (a finished track's lane sits
// idle), the weighted overlap-add and the normalisation run per lane on the device, finished regions are downloaded
// while later segments run.  nt == 1 is umx_hip_split_inference / umx_hip_shift_inference.
int umx_hip_ctx::tracks(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host,
                        unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    // A persistent-kernel timeout inside a track cannot be repaired segment by segment (the overlap-add has consumed
    // the stems): everything is run again, once, with the per-step driver the timeout switches the context to.
    no_recovery = true;
    int rc = tracks_once(nt, audio_host, length, shift_offset, out_host, flags, progress, progress_user);
    if (rc == UMX_ERR_TIMEOUT)
        rc = tracks_once(nt, audio_host, length, shift_offset, out_host, flags, progress, progress_user);
    no_recovery = false;
    pending.clear();
    pending_lost = false;
    return rc;
}

int umx_hip_ctx::tracks_once(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host,
                             unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    if (nt < 1 || nt > B || !audio_host || !length || !shift_offset || !out_host)
    {
        set_error("tracks: need 1 <= n_tracks <= the context's track count and non-null argument arrays");
        return UMX_ERR_ARG;
    }
    for (int ln = 0; ln < nt; ++ln)
        if (!audio_host[ln] || length[ln] < 1 || shift_offset[ln] >= UMX_MAX_SHIFT || !out_host[4 * ln] || !out_host[4 * ln + 1] ||
            !out_host[4 * ln + 2] || !out_host[4 * ln + 3])
        {
            set_error("track: need audio, outputs, length >= 1 and shift offset < 22050");
            return UMX_ERR_ARG;
        }
    if (ph_next != -1)
    {
        set_error("track: a phased segment is open (umx_hip_segment_end first)");
        return UMX_ERR_ARG;
    }
    // Reset mode (UMX_FLAG_RESET_SEGMENTS, SURVEY 8(e)'s mode table): every segment starts from a ZERO lstm_data instead of the one
    // the previous segment left behind (umx.cpp:167-171 makes it once per track, :226-227 hands it to every call) -- a declared
    // deviation, opt-in.  The segments of a track are then independent, and up to B of them ride as the track lanes of ONE call.
    const bool reset_lanes = (flags & UMX_FLAG_RESET_SEGMENTS) != 0;
    flags &= ~(unsigned)UMX_FLAG_RESET_SEGMENTS;
    if (reset_lanes && (nt != 1 || !lstm_batched || B < 2))
    {
        set_error("track: UMX_FLAG_RESET_SEGMENTS takes ONE track on a context made by umx_hip_create_tracks with at least 2 lanes");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    if (int rc = sync_all())
        return rc;
    int lead[LSTMB_MAX_TRACKS], L2[LSTMB_MAX_TRACKS], L2max = 0;
    for (int ln = 0; ln < nt; ++ln)
    {
        lead[ln] = shift_offset[ln] < 0 ? 0 : shift_offset[ln];
        // umx.cpp:120-122: length + max_shift - offset -- which the reference overruns for offset > max_shift / 2 (its
        // block write is [offset, offset + length)); the same size where the reference is defined, large enough elsewhere
        const long long l2 = shift_offset[ln] < 0 ? (long long)length[ln]
                                                  : (long long)length[ln] + std::max(UMX_MAX_SHIFT - shift_offset[ln], shift_offset[ln]);
        if (l2 > 0x7fffffff / 2)
        {
            set_error("track: too long");
            return UMX_ERR_ARG;
        }
        L2[ln] = (int)l2;
        L2max = std::max(L2max, L2[ln]);
    }
    const int stride = (int)((1 - 0.25f) * N); // umx.cpp:181, inference.hpp:15
    // lanes whose per-segment stem buffers are needed: one per track, or (reset mode) one per segment of a call
    const int seg_lanes = reset_lanes ? (int)std::min<long long>(B, ((long long)L2max + stride - 1) / stride) : nt;
    if (trk.size() < (size_t)seg_lanes)
        trk.resize(seg_lanes);
    for (int ln = 0; ln < seg_lanes; ++ln)
    {
        TrackBufs &tb_ = trk[ln];
        if (ln < nt && (size_t)L2[ln] > tb_.cap) // grow-only track buffers
        {
            const size_t cap = (size_t)L2[ln] + (size_t)L2[ln] / 8;
            for (float **p : {&tb_.in, &tb_.out[0], &tb_.out[1], &tb_.out[2], &tb_.out[3], &tb_.sumw})
                if (*p)
                {
                    allocs.erase(std::find(allocs.begin(), allocs.end(), (void *)*p));
                    (void)hipFree(*p);
                    *p = nullptr;
                }
            tb_.cap = 0;
            if (int rc = dalloc(&tb_.in, 2 * cap, false))
                return rc;
            for (int t = 0; t < 4; ++t)
                if (int rc = dalloc(&tb_.out[t], 2 * cap, false))
                    return rc;
            if (int rc = dalloc(&tb_.sumw, cap, false))
                return rc;
            tb_.cap = cap;
        }
        if (!tb_.seg[0][0])
            for (int s = 0; s < nslots; ++s)
                for (int t = 0; t < 4; ++t)
                    if (int rc = dalloc(&tb_.seg[s][t], (size_t)2 * N, false))
                        return rc;
    }
    if (!trk_acc_ev[0])
        for (int s = 0; s < nslots; ++s)
            UMX_HIP_CHECK(hipEventCreateWithFlags(&trk_acc_ev[s], hipEventDisableTiming));
    // umx.cpp:167-171: a fresh, zeroed lstm_data per track; umx.cpp:186-195: zeroed accumulators (and F4)
    UMX_HIP_CHECK(hipMemset(state, 0, sizeof(float) * state_floats() * (reset_lanes ? B : nt)));
    clear_used();
    for (int ln = 0; ln < nt; ++ln) // (reset mode: nt = 1)
    {
        TrackBufs &tb_ = trk[ln];
        UMX_HIP_CHECK(hipMemset(tb_.in, 0, sizeof(float) * 2 * (size_t)L2[ln]));
        for (int t = 0; t < 4; ++t)
            UMX_HIP_CHECK(hipMemset(tb_.out[t], 0, sizeof(float) * 2 * (size_t)L2[ln]));
        UMX_HIP_CHECK(hipMemset(tb_.sumw, 0, sizeof(float) * (size_t)L2[ln]));
    }
    UMX_HIP_CHECK(hipDeviceSynchronize());
    // The track goes up segment by segment (round 6): a call needs the samples up to the end of its last segment, and the rest of a
    // pageable upload (14 GB/s: 15 ms for ten minutes of stereo) runs while the device is busy with the segments before it -- the
    // slots' streams do not wait for the null stream, and launches are queued ahead of the copy.
    long long uploaded[LSTMB_MAX_TRACKS] = {}; // host samples of each track already on the device
    auto upload_until = [&](int ti, long long padded_end) -> hipError_t { // samples of the padded signal below `padded_end` must be there
        const long long want = std::min<long long>(length[ti], padded_end - lead[ti]);
        if (want <= uploaded[ti])
            return hipSuccess;
        const hipError_t e = hipMemcpy(trk[ti].in + 2 * ((size_t)lead[ti] + (size_t)uploaded[ti]), audio_host[ti] + 2 * (size_t)uploaded[ti],
                                       sizeof(float) * 2 * (size_t)(want - uploaded[ti]), hipMemcpyHostToDevice);
        uploaded[ti] = want;
        return e;
    };

    const float total_reps = std::ceil((float)L2max / (float)stride); // umx.cpp:208 (of the longest track)
    float done = 0.f;
    // A sample is final once the segment that starts at or before it and the one before that have been blended
    // in: region [offset_i, offset_{i+1}) right after segment i.  It is normalised there and then, and the host
    // downloads it while the GPU is already busy with the following segments.
    struct Region
    {
        int lane, start, count;
        hipEvent_t ready;
    };
    std::vector<Region> regions;
    auto cleanup = [&]() {
        for (Region &r : regions)
            (void)hipEventDestroy(r.ready);
    };
    int last_slot = -1, iseg = 0;
    const int per_call = reset_lanes ? seg_lanes : 1; // segments of a track per call
    for (long long off = 0; off < L2max; off += (long long)stride * per_call, ++iseg)
    {
        const int si = next_slot();
        const float *ain[LSTMB_MAX_TRACKS] = {};
        int nn[LSTMB_MAX_TRACKS] = {}, seg_off[LSTMB_MAX_TRACKS] = {};
        float *outs[4 * LSTMB_MAX_TRACKS] = {};
        const int nl = reset_lanes ? per_call : nt; // lanes of this call: lane = track (its segment at `off`), or lane k = segment k of the call
        for (int ln = 0; ln < nl; ++ln)
        {
            const int ti = reset_lanes ? 0 : ln;
            const long long so = reset_lanes ? off + (long long)ln * stride : off;
            if (so < L2[ti]) // this track still has a segment here (umx.cpp:214-217)
            {
                seg_off[ln] = (int)so;
                ain[ln] = trk[ti].in + 2 * (size_t)so;
                nn[ln] = std::min(N, L2[ti] - (int)so);
                if (upload_until(ti, so + N) != hipSuccess)
                {
                    cleanup();
                    set_error("track: upload failed");
                    return UMX_ERR_HIP;
                }
                for (int t = 0; t < 4; ++t)
                    outs[4 * ln + t] = trk[ln].seg[si][t];
            }
        }
        if (reset_lanes && iseg > 0) // every segment from a zero state: the previous call has left its own behind
        {
            if (int rc = sync_all())
            {
                cleanup();
                return rc;
            }
            UMX_HIP_CHECK(hipMemset(state, 0, sizeof(float) * state_floats() * B));
            UMX_HIP_CHECK(hipDeviceSynchronize());
        }
        const int rc = infer_batch(nl, ain, nn, outs, flags);
        if (rc)
        {
            cleanup();
            return rc;
        }
        hipStream_t st = slot[si].stream;
        if (last_slot >= 0) // accumulate in segment order (two segments overlap by a quarter)
            (void)hipStreamWaitEvent(st, trk_acc_ev[last_slot], 0);
        for (int ln = 0; ln < nl; ++ln)
        {
            if (!ain[ln])
                continue;
            const int ti = reset_lanes ? 0 : ln, offset = seg_off[ln];
            Stems4 tk, seg;
            for (int t = 0; t < 4; ++t)
            {
                tk.p[t] = reinterpret_cast<float2 *>(trk[ti].out[t]);
                seg.p[t] = reinterpret_cast<float2 *>(trk[ln].seg[si][t]);
            }
            hipLaunchKernelGGL(track_accumulate_kernel, dim3((nn[ln] + 255) / 256, 4), dim3(256), 0, st, tk, trk[ti].sumw, seg, offset, nn[ln], N);
            Region rg;
            rg.lane = ti;
            rg.start = offset;
            rg.count = (int)std::min<long long>((long long)offset + stride, L2[ti]) - offset;
            hipLaunchKernelGGL(track_normalise_kernel, dim3((rg.count + 255) / 256, 4), dim3(256), 0, st, tk, trk[ti].sumw, rg.start, rg.count);
            if (hipEventCreateWithFlags(&rg.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(rg.ready, st) != hipSuccess)
            {
                cleanup();
                set_error("track: event creation failed");
                return UMX_ERR_HIP;
            }
            regions.push_back(rg);
        }
        (void)hipEventRecord(trk_acc_ev[si], st);
        last_slot = si;
        done += 1.0f / total_reps; // umx.cpp:229 (queued, not finished: the device runs behind the host here)
        if (progress)
            progress(done, progress_user);
    }
    hipError_t cerr = hipSuccess;
    for (const Region &rg : regions) // umx.cpp:136-147: drop the shift
    {
        const int ln = rg.lane;
        const long long lo = std::max<long long>(rg.start, lead[ln]),
                        hi = std::min<long long>((long long)rg.start + rg.count, (long long)lead[ln] + length[ln]);
        if (hi <= lo)
            continue;
        if (cerr == hipSuccess)
            cerr = hipEventSynchronize(rg.ready);
        for (int t = 0; t < 4 && cerr == hipSuccess; ++t)
            cerr = hipMemcpy(out_host[4 * ln + t] + 2 * (size_t)(lo - lead[ln]), trk[ln].out[t] + 2 * (size_t)lo,
                             sizeof(float) * 2 * (size_t)(hi - lo), hipMemcpyDeviceToHost);
    }
    cleanup();
    if (cerr != hipSuccess)
    {
        set_error(hipGetErrorString(cerr));
        return UMX_ERR_HIP;
    }
    UMX_HIP_CHECK(hipGetLastError());
    if (int rc = umx_hip_sync(this)) // surfaces a persistent-kernel timeout
        return rc;
    return UMX_OK;
}
