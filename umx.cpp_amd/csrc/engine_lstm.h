// engine_lstm.h -- part of engine.hip (one translation unit: the kernels inline into their launchers): one LSTM layer: the one-track persistent / per-step drivers and the track-batched kernels, through the admission gate.
// Included by engine.hip behind the definition of umx_hip_ctx; not a stand-alone header.
// ---------------------------------------------------------------- LSTM layer
int umx_hip_ctx::run_lstm_layer(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask)
{
    if (lstm_batched)
        return run_lstm_layer_batched(sl, layer, active, nact, stepwise, lane_mask);
    hipStream_t st = sl.stream;
    LstmArgs a;
    memset(&a, 0, sizeof a);
    a.W = whh[layer];      // nullptr when W_hh is u8-resident
    a.Wq = whh_q[layer];
    for (int c = 0; c < 8; ++c)
    {
        a.wsc[c] = whh_s[layer][c];
        a.wof[c] = whh_o[layer][c];
    }
    a.bhh = bhh[layer];
    a.state = state;
    a.hbuf = sl.hbuf;
    a.sync = sl.lsync;
    a.status = sl.status;
    a.prof = (last_flags & UMX_FLAG_LSTM_PROFILE) ? sl.lprof : nullptr;
    a.force_safe = (last_flags & UMX_FLAG_LSTM_FORCE_SAFE) ? 1 : 0;
    a.poll_delay = lstm_poll_delay;
    a.Hl = Hl;
    a.S = S;
    a.T = T;
    a.ldp = 4 * H;
    a.layer = layer;
    for (int i = 0; i < 4; ++i)
    {
        const TargetAct &b = sl.lane[0].ta[i];
        a.P[i] = b.P;
        if (layer == 0)
        {
            a.out[i] = b.la;
            a.ldo = H;
            a.col0 = 0;
        }
        else if (layer == 1)
        {
            a.out[i] = b.lb;
            a.ldo = H;
            a.col0 = 0;
        }
        else
        {
            a.out[i] = b.cat; // inference.cpp:118-123 skip concat: lstm output -> right half of cat
            a.ldo = 2 * H;
            a.col0 = H;
        }
        a.tmap[i] = i < nact ? active[i] : 0;
    }
    const int nchains = 2 * nact;
    a.nchains = nchains;
    const dim3 grid(S, nchains), block(LSTM_THREADS);
    const int kpw = Hl / 8;
    bool persistent = !stepwise && persistent_ok && (kpw == 8 || kpw == 16 || kpw == 32 || kpw == 64) &&
                      8 * S <= lstm_capacity;
    if (persistent)
    {
        a.tag_base = next_tag_base();
        a.abort_at = ((last_flags & UMX_FLAG_DEBUG_LSTM_ABORT) && layer == 1) ? T / 2 : 0;
        // census + arrival counter every launch; the granule area only when the tag epoch wraps
        UMX_HIP_CHECK(hipMemsetAsync(sl.lsync, 0, sizeof(unsigned) * (tag_epoch == 0 ? lsync_words : LSTM_SYNC_HEADER_WORDS), st));
        void *kargs[] = {&a};
        const void *fn = lstm_persistent_fn(kpw, last_flags & UMX_FLAG_PRECISE_ACT);
        // The granule exchange needs the whole grid co-resident.  Residency was checked against the
        // occupancy of this kernel at create time (lstm_capacity); a plain launch is used because ROCm
        // serialises cooperative launches against other queues, which would defeat the two-slot overlap.
        // Every spin in the kernel is bounded, so a grid that is not resident after all ends as
        // UMX_ERR_TIMEOUT, not as a hang.
        // always 8*S workgroups: with round-robin dispatch every XCD then receives S of them and the
        // census can enable the intra-XCD protocol; surplus workgroups (skipped targets) exit at once
        hipError_t e = lstm_gate_launch(device, st, 8 * S * (lstm_threads > 512 ? 2 : 1), 2 * n_cus,
                                        [&] { return hipLaunchKernel(fn, dim3(8 * S), dim3(lstm_threads), kargs, 0, st); });
        if (e != hipSuccess)
        {
            (void)hipGetLastError();
            persistent_ok = false;
            persistent = false;
        }
    }
    if (!persistent)
    {
        hipLaunchKernelGGL(lstm_state_to_hbuf, dim3(nchains), dim3(256), 0, st, a, nchains);
        if (last_flags & UMX_FLAG_PRECISE_ACT)
            for (int step = 0; step < T; ++step)
                hipLaunchKernelGGL(lstm_step_kernel<true>, grid, block, 0, st, a, step);
        else
            for (int step = 0; step < T; ++step)
                hipLaunchKernelGGL(lstm_step_kernel<false>, grid, block, 0, st, a, step);
        hipLaunchKernelGGL(lstm_hbuf_to_state, dim3(nchains), dim3(256), 0, st, a, nchains);
    }
    sl.last_persistent = persistent;
    lstm_kernel_last = persistent ? "lstm_persistent_kernel" : "lstm_step_kernel";
    UMX_HIP_CHECK(hipGetLastError());
    return UMX_OK;
}

// All track lanes of `lane_mask` through ONE launch per layer (lstm_batch.h).  stepwise (or a grid that cannot be
// co-resident): the same kernel one step per launch, carried through the fp32 stream state -- bit-identical.
int umx_hip_ctx::run_lstm_layer_batched(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask)
{
    hipStream_t st = sl.stream;
    LstmBArgs a;
    memset(&a, 0, sizeof a);
    a.W = whh[layer];
    a.Wq = whh_q[layer];
    for (int c = 0; c < 8; ++c)
    {
        a.wsc[c] = whh_s[layer][c];
        a.wof[c] = whh_o[layer][c];
    }
    a.bhh = bhh[layer];
    a.state = state;
    a.state_out = state;
    a.state_stride = state_floats();
    a.sync = sl.lsync;
    a.status = sl.status;
    a.prof = (last_flags & UMX_FLAG_LSTM_PROFILE) ? sl.lprof : nullptr;
    a.force_safe = (last_flags & UMX_FLAG_LSTM_FORCE_SAFE) ? 1 : 0;
    a.Hl = Hl;
    a.S = S;
    a.T = T;
    a.ldp = 4 * H;
    a.layer = layer;
    a.p_stride = (size_t)Tp * 4 * H;
    a.out_stride = layer == 2 ? (size_t)Tp * 2 * H : (size_t)Tp * H;
    a.ldo = layer == 2 ? 2 * H : H;
    a.col0 = layer == 2 ? H : 0; // inference.cpp:118-123 skip concat: lstm output -> right half of cat
    for (int i = 0; i < 4; ++i)
    {
        const TargetAct &b = sl.lane[0].ta[i];
        a.P[i] = b.P;
        a.out[i] = layer == 0 ? b.la : layer == 1 ? b.lb : b.cat;
        a.tmap[i] = i < nact ? active[i] : 0;
    }
    a.nchains = 2 * nact;
    a.lane_mask = lane_mask;
    a.poll_delay = lstm8_poll_delay;
    const bool wq_layer = whh_q[layer] != nullptr && !u8_dequant;
    const bool fuse = lstm_rowsums && wq_layer;
    if (fuse)
    {
        const size_t rows_all = (size_t)B * Tp + Mpad;
        for (int i = 0; i < 4; ++i)
        {
            const TargetAct &b = sl.lane[0].ta[i];
            a.planes[i] = layer == 0 ? b.la_p : layer == 1 ? b.lb_p : b.cat_p; // (dropped below if the launch's kernel cannot write them)
            a.rs_dir[i] = layer == 0 ? b.rs_la : layer == 1 ? b.rs_lb : b.rs_catR;
        }
        a.ldpl = a.ldo;
        a.plane_elems = rows_all * (size_t)a.ldo;
        a.rs_rows = rows_all;
    }
    a.Tp = Tp;
    int top = 0;
    for (int ln = 0; ln < LSTMB_MAX_TRACKS; ++ln)
        if ((lane_mask >> ln) & 1ull)
            top = ln + 1;
    const bool wq = whh_q[layer] != nullptr && !u8_dequant;
    // ONE recurrence per context shape (round 6).  lstm_batch8.h (LSTM hidden 512 / 256, u8-resident W_hh): a launch serves the octets
    // the chip holds side by side -- 32 lanes (hidden 1024) or 64 (hidden 512) --, twice that with two octets per workgroup in turn.
    // Everything else: lstm_batch_kernel, a group of 16 lanes per launch.  More lanes than a launch serves: the parts one after the other
    // (each a complete layer of its lanes: per (unit, lane) the arithmetic does not know who else is in the context).
    const bool octets = lstm_batch8_ok;
    const int per = octets ? LSTM8_TRACKS * lstm8_octets(Hl) : LSTMB_GROUP_TRACKS;
    const int octs = octets && env_lstm8_paired && top > per ? 2 : 1; // octets per workgroup, in turn (UMX_LSTM8_PAIRED=0: two launches instead)
    const int span = per * octs, parts = (top + span - 1) / span;
    a.nbp = octets ? 16 : std::min(top, span) > 8 ? 16 : std::min(top, span) > 4 ? 8 : std::min(top, span) > 2 ? 4 : std::min(top, span) > 1 ? 2 : 1;
    a.bulk = a.nbp > 8 ? 8 : 16;
    const size_t lds = octets ? lstm8_lds_bytes(Hl, octs) : lstmb_lds_bytes(a.nbp, a.bulk);
    const void *fn = octets ? lstm_batch8_fn(Hl, last_flags & UMX_FLAG_PRECISE_ACT, octs) : lstm_batch_fn(Hl, wq, last_flags & UMX_FLAG_PRECISE_ACT);
    lstm_kernel_last = octets ? "lstm_batch8_kernel" : "lstm_batch_kernel";
    const int threads = LSTM_THREADS;
    const int Sw = octets ? 32 : S; // workgroups per chain of the launch's grid
    const bool writes_planes = fuse && lstm_writes_planes;
    if (!writes_planes)
        for (int i = 0; i < 4; ++i)
            a.planes[i] = nullptr;
    a.write_f32 = (last_flags & UMX_FLAG_DEBUG_TAPS) ? 1 : 0;
    sl.lstm_wrote_planes[layer] = writes_planes;
    sl.lstm_rows_f32[layer] = !writes_planes || a.write_f32;
    void *kargs[] = {&a};
    auto part_on = [&](int p) {
        const unsigned long long m = span >= 64 ? ~0ull : ((1ull << span) - 1ull);
        return ((lane_mask >> (span * p)) & m) != 0;
    };
    bool persistent = !stepwise && persistent_ok && 8 * S <= lstm_batch_capacity;
    int queued = 0;
    for (int p = 0; persistent && p < parts; ++p)
    {
        if (!part_on(p))
            continue;
        a.lane_base = span * p;
        a.tag_epoch = next_tag_base() >> 12; // unique per launch (20 bits); the granule area is cleared when it wraps
        const bool clear = tag_epoch == 0;
        a.t_begin = 0;
        a.t_end = T;
        a.census = 1;
        a.abort_at = ((last_flags & UMX_FLAG_DEBUG_LSTM_ABORT) && layer == 1) ? T / 2 : 0;
        UMX_HIP_CHECK(hipMemsetAsync(sl.lsync, 0, sizeof(unsigned) * (clear ? lsync_words : LSTM_SYNC_HEADER_WORDS), st));
        hipError_t e = lstm_gate_launch(device, st, 8 * S * 2, 2 * n_cus,
                                        [&] { return hipLaunchKernel(fn, dim3(8 * Sw), dim3(threads), kargs, lds, st); });
        if (e != hipSuccess)
        {
            (void)hipGetLastError();
            persistent_ok = false;
            persistent = false;
            if (queued > 0) // an earlier part of this layer is already on the stream: re-running the layer per step would advance its lanes twice
            {
                set_error(std::string("batched LSTM: launch of a later part of a layer refused: ") + hipGetErrorString(e));
                return UMX_ERR_HIP;
            }
        }
        else
            ++queued;
    }
    if (!persistent)
    {
        a.census = 0;
        a.abort_at = 0;
        // One step per launch, h / c carried through the fp32 stream state.  Every workgroup reads the h of its whole chain at
        // the start of a launch and writes its own units at the end, so a launch must not update the copy it reads: the
        // workgroups of a grid do not start together (round 3: one GPU-suite run in nine failed the per-step bitwise test at
        // 20 lanes; rounds 2-3 updated `state` in place).  This layer's entries ping-pong between `state` and `state_alt`
        // (rows = lanes x targets, 4 Hl floats of every 12 Hl: the other layers' entries may be in use by another slot).
        const size_t row = (size_t)4 * Hl * sizeof(float), pitch = 3 * row, off = (size_t)layer * 4 * Hl;
        UMX_HIP_CHECK(hipMemcpy2DAsync(state_alt + off, pitch, state + off, pitch, row, (size_t)4 * B, hipMemcpyDeviceToDevice, st));
        for (int step = 0; step < T; ++step)
        {
            a.t_begin = step;
            a.t_end = step + 1;
            a.state = (step & 1) ? state_alt : state;
            a.state_out = (step & 1) ? state : state_alt;
            for (int p = 0; p < parts; ++p)
                if (part_on(p))
                {
                    a.lane_base = span * p;
                    UMX_HIP_CHECK(hipLaunchKernel(fn, dim3(2 * nact * Sw), dim3(threads), kargs, lds, st));
                }
        }
        if (T & 1) // the last launch wrote state_alt
            UMX_HIP_CHECK(hipMemcpy2DAsync(state + off, pitch, state_alt + off, pitch, row, (size_t)4 * B, hipMemcpyDeviceToDevice, st));
    }
    if (fuse) // the one row per direction that no later step multiplied with
        hipLaunchKernelGGL(lstm_last_row_sum_kernel, dim3(top, 2 * nact), dim3(64), 0, st, a, nact);
    sl.last_persistent = persistent;
    UMX_HIP_CHECK(hipGetLastError());
    return UMX_OK;
}
