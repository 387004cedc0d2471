// engine_stages.h -- part of engine.hip (one translation unit: the kernels inline into their launchers): the stages of one segment (GEMM launches, STFT front, masks, Wiener / inverse STFT back) and umx_inference for every track lane.
// Included by engine.hip behind the definition of umx_hip_ctx; not a stand-alone header.
// ---------------------------------------------------------------- stages of one segment
void umx_hip_ctx::launch_gemm(Lane &sl, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg)
{
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.M = Tp;
    g.T = T;
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const TargetAct &c = sl.ta[active[i]];
        GemmTarget &t = g.t[i];
        switch (mode)
        {
        case G_FC1:
            t.A = sl.x; t.B = b.fc1_w; t.C = c.cat;
            t.e0 = b.bn1[0]; t.e1 = b.bn1[1]; t.e2 = b.bn1[2]; t.e3 = b.bn1[3];
            t.q0 = b.in_scale; t.q1 = b.in_mean;
            g.N = H; g.K = KX; g.lda = KX; g.ldc = 2 * H;
            break;
        case G_IH:
            t.A = layer == 0 ? c.cat : layer == 1 ? c.la : c.lb;
            t.B = b.ih_w[layer]; t.C = c.P; t.e0 = b.ih_b[layer];
            g.N = 4 * H; g.K = H; g.lda = layer == 0 ? 2 * H : H; g.ldc = 4 * H;
            break;
        case G_FC2:
            t.A = c.cat; t.B = b.fc2_w; t.C = c.a2;
            t.e0 = b.bn2[0]; t.e1 = b.bn2[1]; t.e2 = b.bn2[2]; t.e3 = b.bn2[3];
            g.N = H; g.K = 2 * H; g.lda = 2 * H; g.ldc = H;
            break;
        default:
            t.A = c.a2; t.B = b.fc3_w; t.C = c.mag;
            t.e0 = b.bn3[0]; t.e1 = b.bn3[1]; t.e2 = b.bn3[2]; t.e3 = b.bn3[3];
            t.q0 = b.out_scale; t.q1 = b.out_mean;
            g.N = NOUT_PAD; g.K = H; g.lda = H; g.ldc = 0;
            break;
        }
    }
    // quantised-resident B (config 5): all active targets were loaded the same way
    int bq = BQ_F32;
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const QMat &q = mode == G_FC1 ? b.fc1_q : mode == G_IH ? b.ih_q[layer] : mode == G_FC2 ? b.fc2_q : b.fc3_q;
        GemmTarget &t = g.t[i];
        t.bsplit = mode == G_IH ? 2 * H : 0x7fffffff; // W_ih rows >= 4*Hl belong to the reverse direction's tensor
        t.bs[0] = t.bs[1] = 1.f;
        if (q.q)
        {
            t.Bq = q.q;
            t.bs[0] = q.s[0]; t.bs[1] = q.s[1];
            t.bo[0] = q.o[0]; t.bo[1] = q.o[1];
            bq = q.type;
        }
    }
    if (gemm_bf16x3 && bq == BQ_F32) // weights resident as three bf16 planes
        for (int i = 0; i < nact; ++i)
        {
            const TargetBufs &b = tb[active[i]];
            g.t[i].Bq = mode == G_FC1 ? b.fc1_bx : mode == G_IH ? b.ih_bx[layer] : mode == G_FC2 ? b.fc2_bx : b.fc3_bx;
        }
    const dim3 grid((unsigned)round_up((g.N / GEMM_BN) * (g.M / GEMM_BM), 8), 1, nact), block(256);
    gemm_kernel_last[mode] = "gemm_bf16x3_kernel";
#define UMX_LAUNCH(KERNEL, LDS) hipLaunchKernelGGL((KERNEL), grid, block, LDS, st, g)
    switch (mode)
        {
        case G_FC1:
            if (bq == BQ_U8 && !u8_dequant) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC1, BQ_U8X>), BX_LDS_BYTES);
            else if (bq == BQ_U8) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC1, BQ_U8>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_FC1, BQ_F32>), BX_LDS_BYTES);
            break;
        case G_IH:
            if (bq == BQ_U8 && !u8_dequant) UMX_LAUNCH((gemm_bf16x3_kernel<G_IH, BQ_U8X>), BX_LDS_BYTES);
            else if (bq == BQ_U8) UMX_LAUNCH((gemm_bf16x3_kernel<G_IH, BQ_U8>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_IH, BQ_F32>), BX_LDS_BYTES);
            break;
        case G_FC2:
            if (bq == BQ_U16) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC2, BQ_U16>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_FC2, BQ_F32>), BX_LDS_BYTES);
            break;
        default:
            if (bq == BQ_U16) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC3, BQ_U16>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_FC3, BQ_F32>), BX_LDS_BYTES);
            break;
        }
#undef UMX_LAUNCH
}

// gemm_planes.h: split one A operand of every active target into bf16 planes + row sums
// ln = the first of nl consecutive track lanes (their buffers are contiguous: see init)
void umx_hip_ctx::launch_split(Lane &ln, int nl, hipStream_t st, int which, const int *active, int nact)
{
    SplitArgs a;
    memset(&a, 0, sizeof a);
    a.T = T;
    a.Tp = Tp;
    const size_t rows_all = (size_t)B * Tp + Mpad; // rows of one output plane
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const TargetAct &c = ln.ta[active[i]];
        switch (which)
        {
        case SP_XS: // x * input_scale + input_mean (inference.cpp:78-83), per target
            a.src[i] = ln.x; a.dst[i] = c.xs_p; a.rowsum[i] = c.rs_xs; a.rowunscale[i] = c.rsc_xs; a.scale[i] = b.in_scale; a.mean[i] = b.in_mean;
            a.cols = KX; a.ld_src = KX; a.ld_dst = KX; a.col0_dst = 0; a.plane = rows_all * KX;
            break;
        case SP_CATL: // fc1 output = left half of the skip concat
            a.src[i] = c.cat; a.dst[i] = c.cat_p; a.rowsum[i] = c.rs_catL;
            a.cols = H; a.ld_src = 2 * H; a.ld_dst = 2 * H; a.col0_dst = 0; a.plane = rows_all * 2 * H;
            break;
        case SP_CATR: // last LSTM layer's output = right half
            a.src[i] = c.cat + H; a.dst[i] = c.cat_p; a.rowsum[i] = lstm_rowsums ? nullptr : c.rs_catR;
            a.cols = H; a.ld_src = 2 * H; a.ld_dst = 2 * H; a.col0_dst = H; a.plane = rows_all * 2 * H;
            break;
        case SP_LA:
            a.src[i] = c.la; a.dst[i] = c.la_p; a.rowsum[i] = lstm_rowsums ? nullptr : c.rs_la;
            a.cols = H; a.ld_src = H; a.ld_dst = H; a.col0_dst = 0; a.plane = rows_all * H;
            break;
        case SP_LB:
            a.src[i] = c.lb; a.dst[i] = c.lb_p; a.rowsum[i] = lstm_rowsums ? nullptr : c.rs_lb;
            a.cols = H; a.ld_src = H; a.ld_dst = H; a.col0_dst = 0; a.plane = rows_all * H;
            break;
        default:
            a.src[i] = c.a2; a.dst[i] = c.a2_p; a.rowsum[i] = c.rs_a2; a.rowunscale[i] = c.rsc_a2;
            a.cols = H; a.ld_src = H; a.ld_dst = H; a.col0_dst = 0; a.plane = rows_all * H;
            break;
        }
    }
    a.rows_valid = nl * Tp;
    const dim3 grid(round_up(nl * Tp, 256) / 4, 1, nact);
    if (which == SP_XS && a.cols <= 3072) // every target's operand comes from the same rows of x: read them once
    {
        hipLaunchKernelGGL(split_planes_shared_kernel<6>, dim3(grid.x), dim3(256), 0, st, a, nact);
        return;
    }
    if (a.cols <= 1024)
        hipLaunchKernelGGL(split_planes_kernel<2>, grid, dim3(256), 0, st, a);
    else if (a.cols <= 3072)
        hipLaunchKernelGGL(split_planes_kernel<6>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(split_planes_kernel<8>, grid, dim3(256), 0, st, a);
}

void umx_hip_ctx::launch_gemm_planes(Lane &ln, int nl, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg)
{
    GemmPArgs g;
    memset(&g, 0, sizeof g);
    g.M = round_up(nl * Tp, 256); // rows behind the last lane of the launch are zero planes (split_planes_kernel)
    g.lanes = nl;
    g.T = T;
    g.Tp_lane = Tp;
    g.mag_lane = (size_t)2 * T * MAGP;
    g.a_unscale = 1.0f / (float)(1 << GP_SPLIT_FIXED_EXP); // tanh / LSTM outputs: constant scale (split_planes_kernel)
    const size_t rows_all = (size_t)B * Tp + Mpad;
    int nbp = 2;
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const TargetAct &c = ln.ta[active[i]];
        GemmPTarget &t = g.t[i];
        const PMat *pm = nullptr;
        t.bsplit = 0x7fffffff;
        switch (mode)
        {
        case G_FC1:
            pm = &b.fc1_p;
            t.A = c.xs_p; t.C = c.cat; t.rs0 = c.rs_xs; t.rsc = c.rsc_xs;
            t.e0 = b.bn1[0]; t.e1 = b.bn1[1]; t.e2 = b.bn1[2]; t.e3 = b.bn1[3];
            g.N = H; g.K = KX; g.lda = KX; g.ldc = 2 * H; g.a_plane = rows_all * KX;
            break;
        case G_IH:
            pm = &b.ih_p[layer];
            t.A = layer == 0 ? c.cat_p : layer == 1 ? c.la_p : c.lb_p;
            t.rs0 = layer == 0 ? c.rs_catL : layer == 1 ? c.rs_la : c.rs_lb;
            if (layer > 0 && lstm_rowsums) // one row-sum array per direction, from the recurrence itself
                t.rs1 = t.rs0 + rows_all;
            t.C = c.P; t.e0 = b.ih_b[layer];
            t.bsplit = 2 * H; // W_ih rows >= 4*Hl belong to the reverse direction's tensor
            g.N = 4 * H; g.K = H; g.lda = layer == 0 ? 2 * H : H; g.ldc = 4 * H;
            g.a_plane = layer == 0 ? rows_all * 2 * H : rows_all * H;
            break;
        case G_FC2:
            pm = &b.fc2_p;
            t.A = c.cat_p; t.C = c.a2; t.rs0 = c.rs_catL; t.rs1 = c.rs_catR;
            if (lstm_rowsums)
                t.rs2 = c.rs_catR + rows_all;
            t.e0 = b.bn2[0]; t.e1 = b.bn2[1]; t.e2 = b.bn2[2]; t.e3 = b.bn2[3];
            g.N = H; g.K = 2 * H; g.lda = 2 * H; g.ldc = H; g.a_plane = rows_all * 2 * H;
            break;
        default:
            pm = &b.fc3_p;
            t.A = c.a2_p; t.C = c.mag; t.rs0 = c.rs_a2; t.rsc = c.rsc_a2;
            t.e0 = b.bn3[0]; t.e1 = b.bn3[1]; t.e2 = b.bn3[2]; t.e3 = b.bn3[3];
            t.q0 = b.out_scale; t.q1 = b.out_mean;
            g.N = NOUT_PAD; g.K = H; g.lda = H; g.ldc = 0; g.a_plane = rows_all * H;
            break;
        }
        t.B = pm->p;
        t.bs[0] = pm->s[0]; t.bs[1] = pm->s[1];
        t.bo2[0] = pm->o2[0]; t.bo2[1] = pm->o2[1];
        nbp = pm->nbp; // the same for every target (all_q at create)
    }
    // 256 x 256 tiles (half the L2 traffic per flop) when they fill the chip
    const int blocks_big = (g.N / 256) * (g.M / 256) * nact;
    const bool big = g.N % 256 == 0 && blocks_big >= 224;
    const int bm = big ? 256 : 128;
    const dim3 grid((unsigned)round_up((g.N / bm) * (g.M / bm), 8), 1, nact), block(big ? 1024 : 256);
    const size_t lds = big ? gp_lds_bytes(4, 4, nbp) : gp_lds_bytes(2, 2, nbp);
    // 256 x 256 blocks: eight waves of 128 x 64 in ping-pong (gemm_planes_pp.h), or sixteen waves of 64 x 64 in lock step
    // (gemm_planes.h: same bits; UMX_GEMM_PP=0, or a bit per GemmMode).  Measured alone, 32 lanes, ms per launch incl. the split
    // kernel, A/B on one box (round 3): fc1 5.70-5.98 -> 5.38-5.52, W_ih 5.69-6.14 -> 5.29-5.67, fc2 4.77 -> 4.62-4.66, fc3 9.21 -> 8.98-9.15.
    const int gemm_pp = env_gemm_pp; // UMX_GEMM_PP, read when the context was created
    const bool pp = big && (gemm_pp < 0 || ((gemm_pp >> mode) & 1));
    // ... as a persistent kernel (gemm_planes_ps.h: one workgroup per CU walks the tiles of all targets, a tile's epilogue inside the
    // next tile's first trip; same bits): launches with more tiles than CUs, UMX_GEMM_PS = bit per GemmMode (0: never)
    const int ps_wgs = n_cus / 8 * 8;
    const bool ps = pp && (env_gemm_ps < 0 || ((env_gemm_ps >> mode) & 1)) && g.K / GP_BK >= 6 && ps_wgs >= 8 && blocks_big > ps_wgs && g.Tp_lane >= 256;
    gemm_kernel_last[mode] = ps ? "gemm_planes_ps_kernel" : pp ? "gemm_planes_pp_kernel" : "gemm_planes_kernel";
#define UMX_GP(MODE)                                                                                                 \
    if (ps && nbp == 1) hipLaunchKernelGGL((gemm_planes_ps_kernel<MODE, 1>), dim3(ps_wgs), dim3(512), ps_lds_bytes(1), st, g, nact); \
    else if (ps) hipLaunchKernelGGL((gemm_planes_ps_kernel<MODE, 2>), dim3(ps_wgs), dim3(512), ps_lds_bytes(2), st, g, nact); \
    else if (pp && nbp == 1) hipLaunchKernelGGL((gemm_planes_pp_kernel<MODE, 1>), grid, dim3(512), lds, st, g);           \
    else if (pp) hipLaunchKernelGGL((gemm_planes_pp_kernel<MODE, 2>), grid, dim3(512), lds, st, g);                  \
    else if (big && nbp == 1) hipLaunchKernelGGL((gemm_planes_kernel<MODE, 1, 4, 4>), grid, block, lds, st, g);      \
    else if (big) hipLaunchKernelGGL((gemm_planes_kernel<MODE, 2, 4, 4>), grid, block, lds, st, g);                  \
    else if (nbp == 1) hipLaunchKernelGGL((gemm_planes_kernel<MODE, 1, 2, 2>), grid, block, lds, st, g);             \
    else hipLaunchKernelGGL((gemm_planes_kernel<MODE, 2, 2, 2>), grid, block, lds, st, g);
    switch (mode)
    {
    case G_FC1: UMX_GP(G_FC1) break;
    case G_IH: UMX_GP(G_IH) break;
    case G_FC2: UMX_GP(G_FC2) break;
    default: UMX_GP(G_FC3) break;
    }
#undef UMX_GP
}

void umx_hip_ctx::launch_gemm_lanes(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, int mode, int layer,
                                    const int *active, int nact, bool dbg)
{
    if (nact <= 0)
        return;
    for (int l0 = 0; l0 < nb;)
    {
        if (!audio_dev[l0])
        {
            ++l0;
            continue;
        }
        int l1 = l0 + 1;
        while (gemm_planes && l1 < nb && audio_dev[l1])
            ++l1;
        if (gemm_planes)
        {
            // the A operand is split here, right before its consumer (every producer -- STFT, fc1, the LSTM layers,
            // fc2 -- writes fp32)
            const int which = mode == G_FC1 ? SP_XS : mode == G_IH ? (layer == 0 ? SP_CATL : layer == 1 ? SP_LA : SP_LB) : mode == G_FC2 ? SP_CATR : SP_A2;
            const int from_layer = which == SP_LA ? 0 : which == SP_LB ? 1 : which == SP_CATR ? 2 : -1;
            if (from_layer < 0 || !sl.lstm_wrote_planes[from_layer]) // else: written by the recurrence
                launch_split(sl.lane[l0], l1 - l0, st, which, active, nact);
            {
                // the stage's kernel time without its split kernel (umx_hip_stage_kernel_times): stage event ... this event ... next stage event
                const int stg = mode == G_FC1 ? ST_FC1 : mode == G_IH ? ST_IH0 + 2 * layer : mode == G_FC2 ? ST_FC2 : ST_FC3;
                sl.evk_set[stg] = sl.evk[stg] && hipEventRecord(sl.evk[stg], st) == hipSuccess;
            }
            launch_gemm_planes(sl.lane[l0], l1 - l0, st, mode, layer, active, nact, dbg);
        }
        else
            launch_gemm(sl.lane[l0], st, mode, layer, active, nact, dbg);
        l0 = l1;
    }
}

// stft -> |.|, crop/stack -> fc1/bn1/tanh -> input projection of LSTM layer 0; stage by stage over the track lanes
int umx_hip_ctx::stage_front(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, const int *n, const int *active,
                             int nact)
{
    stage_range(ST_STFT);
    for (int i = 0; i < ST_COUNT; ++i) // (a stage's kernel-only event counts for the call that recorded it: umx_hip_stage_kernel_times_slot)
        sl.evk_set[i] = false;
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_STFT], st));
    {
        StftIn in;
        in.lanes = lane_set(nb, audio_dev);
        for (int i = 0; i < in.lanes.count; ++i)
        {
            in.audio[i] = audio_dev[in.lanes.id[i]];
            in.n[i] = n[in.lanes.id[i]];
        }
        Lane &L0 = sl.lane[0];
        UMX_HIP_CHECK(hipMemsetAsync(L0.maxabs, 0, sizeof(unsigned) * B, st)); // per-call scratch of every lane
        hipLaunchKernelGGL(stft_kernel, dim3((T + STFT_RUN - 1) / STFT_RUN, in.lanes.count), dim3(256), 0, st, in, N, T, window, tw1, tw2, L0.spec, lane_strides().spec, L0.x,
                           (size_t)Tp * KX, L0.maxabs);
    }
    stage_range(ST_FC1);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_FC1], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_FC1, 0, active, nact, false);
    stage_range(ST_IH0);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_IH0], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_IH, 0, active, nact, false);
    return UMX_OK;
}

// fc2/bn2/relu -> fc3/bn3/scale/relu/mask -> Wiener (or mix phase) -> iSTFT -> overlap-add
int umx_hip_ctx::stage_back(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n,
                            unsigned flags, const int *active, int nact)
{
    if (int rc = stage_masks(sl, st, nb, audio_dev, flags, active, nact))
        return rc;
    return stage_finish(sl, st, nb, audio_dev, out, n, flags, true);
}

int umx_hip_ctx::stage_masks(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, unsigned flags, const int *active, int nact)
{
    const bool dbg = flags & UMX_FLAG_DEBUG_TAPS;
    stage_range(ST_FC2);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_FC2], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_FC2, 0, active, nact, dbg);
    stage_range(ST_FC3);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_FC3], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_FC3, 0, active, nact, dbg);
    UMX_HIP_CHECK(hipGetLastError());
    return UMX_OK;
}

int umx_hip_ctx::stage_finish(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n,
                              unsigned flags, bool zero_skipped)
{
    const bool dbg = flags & UMX_FLAG_DEBUG_TAPS;
    if (zero_skipped)
        for (int ln = 0; ln < nb; ++ln)
            if (audio_dev[ln])
                for (int tg = 0; tg < 4; ++tg) // a skipped target contributes an all-zero magnitude
                    if (flags & UMX_FLAG_SKIP_TARGET(tg))
                        UMX_HIP_CHECK(hipMemsetAsync(sl.lane[ln].ta[tg].mag, 0, sizeof(float) * 2 * T * MAGP, st));
    stage_range(ST_WIENER);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_WIENER], st));
    const int bt = (NBINS + 255) / 256;
    const LaneSet lanes = lane_set(nb, audio_dev);
    const WienerStrides ls = lane_strides();
    Lane &L0 = sl.lane[0]; // the batched kernels take lane 0's pointers and step by ls
    WienerMags wm0;
    for (int s = 0; s < 4; ++s)
        wm0.m[s] = L0.ta[s].mag;
    const int nchunk = (T + WIENER_CHUNK - 1) / WIENER_CHUNK;
    if (flags & UMX_FLAG_NO_WIENER)
    {
        if (!wiener_fused)
            for (int i = 0; i < lanes.count; ++i)
            {
                Lane &L = sl.lane[lanes.id[i]];
                WienerMags wm;
                for (int s = 0; s < 4; ++s)
                    wm.m[s] = L.ta[s].mag;
                const size_t nel = (size_t)2 * T * NBINS;
                hipLaunchKernelGGL(mixphase_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, L.spec, wm, T, L.y);
            }
    }
    else
    {
        // all four sources per thread: the mixture and its phasor are read / formed once (one launch over all lanes, 32 lanes:
        // 2 per thread 2.10 ms, 4 per thread 1.56 ms; lane-by-lane launches of rounds 1-2 had measured 4 per thread slower)
        hipLaunchKernelGGL(wiener_stats4_kernel<WIENER_STATS_NS>, dim3((NBINS + 63) / 64, nchunk * lanes.count, 4 / WIENER_STATS_NS), dim3(64), 0, st, L0.spec, wm0, T, L0.maxabs, L0.wpart, lanes, ls);
        hipLaunchKernelGGL(wiener_finish4_kernel, dim3(bt, 4, lanes.count), dim3(256), 0, st, L0.wpart, T, L0.Rc, lanes, ls);
        if (!wiener_fused)
            for (int i = 0; i < lanes.count; ++i)
            {
                Lane &L = sl.lane[lanes.id[i]];
                WienerMags wm;
                for (int s = 0; s < 4; ++s)
                    wm.m[s] = L.ta[s].mag;
                hipLaunchKernelGGL(wiener_apply_kernel, dim3(bt, T), dim3(256), 0, st, L.spec, wm, T, L.maxabs, L.Rc, L.y);
            }
    }
    OlaOut oo;
    oo.lanes = lanes;
    int nmax = 1;
    for (int i = 0; i < lanes.count; ++i)
    {
        const int ln = lanes.id[i];
        for (int s = 0; s < 4; ++s)
            oo.p[i][s] = out[4 * ln + s];
        oo.n[i] = n[ln];
        nmax = std::max(nmax, n[ln]);
    }
    stage_range(ST_ISTFT);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_ISTFT], st));
    if (!wiener_fused)
    {
        for (int i = 0; i < lanes.count; ++i)
        {
            Lane &L = sl.lane[lanes.id[i]];
            hipLaunchKernelGGL(istft_frames_kernel, dim3(T, 4), dim3(256), 0, st, L.y, T, window, nw, tw1, tw2, L.frames);
        }
    }
    else
    {
        // gains + filter + inverse STFT frame + overlap-add in one pass (wiener_istft.h); y reaches HBM only for the debug
        // tap, the frames only at the seams between the runs of frames the workgroups take
        float2 *ydbg = dbg ? L0.y : nullptr;
        if (sl.out_free_valid) // the stems this slot wrote two calls ago are still being downloaded from the same buffers
            UMX_HIP_CHECK(hipStreamWaitEvent(st, sl.out_free, 0));
        // runs: a few rounds of workgroups over the chip (one workgroup per CU), at least three frames each
        const int runs = std::max(1, std::min(T / 8, (4 * n_cus + lanes.count - 1) / lanes.count));
        const int run_len = std::max(3, (T + runs - 1) / runs), nruns = (T + run_len - 1) / run_len;
        // one 1024-thread workgroup per run, all four sources (two / one source per workgroup, i.e. more workgroups per CU that
        // each repeat the source-independent part, measured 1.7x / 2.7x slower in round 2)
        if (flags & UMX_FLAG_NO_WIENER)
            hipLaunchKernelGGL((wiener_istft_kernel<false>), dim3(nruns, 1, lanes.count), dim3(1024), WI_LDS_BYTES, st, L0.spec, wm0,
                               T, L0.maxabs, L0.Rc, window, nw, tw1, tw2, L0.frames, ydbg, ls, run_len, oo);
        else
            hipLaunchKernelGGL((wiener_istft_kernel<true>), dim3(nruns, 1, lanes.count), dim3(1024), WI_LDS_BYTES, st, L0.spec, wm0,
                               T, L0.maxabs, L0.Rc, window, nw, tw1, tw2, L0.frames, ydbg, ls, run_len, oo);
        stage_range(ST_OLA);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_OLA], st));
        hipLaunchKernelGGL(wiener_ola_edges_kernel, dim3(3 * HOP / 256, nruns * 4, lanes.count), dim3(256), 0, st, L0.frames, ls.frames, T, run_len, oo);
    }
    if (!wiener_fused)
    {
        stage_range(ST_OLA);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_OLA], st));
        if (sl.out_free_valid) // the stems this slot wrote two calls ago are still being downloaded from the same buffers
            UMX_HIP_CHECK(hipStreamWaitEvent(st, sl.out_free, 0));
        hipLaunchKernelGGL(istft_ola_kernel, dim3((nmax + 255) / 256, 4, lanes.count), dim3(256), 0, st, L0.frames, ls.frames, T, oo);
    }
    stage_range(-1);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_COUNT], st));
    UMX_HIP_CHECK(hipGetLastError());
    sl.have_times = true;
    return UMX_OK;
}

// ---------------------------------------------------------------- one segment (of every track lane)
int umx_hip_ctx::infer_device(const float *audio_dev, int n, float *const out[4], unsigned flags)
{
    if (!audio_dev)
    {
        set_error("infer_segment: need 1 <= n <= segment_samples and non-null audio");
        return UMX_ERR_ARG;
    }
    return infer_batch(1, &audio_dev, &n, out, flags);
}

int umx_hip_ctx::infer_batch(int nb, const float *const *audio_dev, const int *n, float *const *out, unsigned flags)
{
    StageRangeCloser close_ranges_on_return;
    if (nb < 1 || nb > B || !audio_dev || !n || !out)
    {
        set_error("infer: need 1 <= n_tracks <= the context's track count and non-null argument arrays");
        return UMX_ERR_ARG;
    }
    unsigned long long lane_mask = 0;
    for (int ln = 0; ln < nb; ++ln)
    {
        if (!audio_dev[ln]) // idle lane: its stream state stays as it is
            continue;
        if (n[ln] < 1 || n[ln] > N)
        {
            set_error("infer_segment: need 1 <= n <= segment_samples and non-null audio");
            return UMX_ERR_ARG;
        }
        for (int s = 0; s < 4; ++s)
            if (!out[4 * ln + s])
            {
                set_error("infer_segment: null output pointer");
                return UMX_ERR_ARG;
            }
        lane_mask |= 1ull << ln;
    }
    if (!lane_mask)
    {
        set_error("infer: no active track lane");
        return UMX_ERR_ARG;
    }
    if (ph_next != -1)
    {
        set_error("infer_segment: a phased segment is open (umx_hip_segment_end first)");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    // Consecutive calls alternate between two slots/streams; a slot is reused two calls later (stream order
    // protects its buffers).  Everything that touches the streaming LSTM state is ordered by events: R_l of
    // this segment waits for R_l of the previous one.
    const int si = next_slot();
    Slot &sl = slot[si];
    Slot &prev = slot[(si + nslots - 1) % nslots];
    hipStream_t st = sl.stream;
    int active[4], nact;
    active_list(flags, active, nact);
    last_flags = flags;
    const size_t call_idx = pending_lost ? (size_t)kBackupCalls : pending.size();
    if (call_idx >= (size_t)kBackupCalls)
    {
        pending_lost = true; // no state backup left for this call: a timeout before the next sync cannot be repaired
        pending.clear();
    }
    else
    {
        PendingCall pc;
        memset(&pc, 0, sizeof pc);
        pc.nb = nb;
        pc.flags = flags;
        for (int ln = 0; ln < nb; ++ln)
        {
            pc.audio[ln] = audio_dev[ln];
            pc.n[ln] = n[ln];
            for (int s2 = 0; s2 < 4; ++s2)
                pc.out[4 * ln + s2] = out[4 * ln + s2];
        }
        pending.push_back(pc);
    }
    // Track-batched contexts: the kernels of consecutive calls run one after the other.  Their workgroups take whole CUs (plane
    // GEMM, batched LSTM, fused Wiener kernel), so kernels of two calls side by side only wait for each other's CUs: since the
    // streaming kernels cover all lanes in one launch the serial step is the faster one (32 lanes: 75.4 against 76.3 ms).  The two
    // slots remain for the buffers: uploads and downloads of neighbouring calls still overlap these kernels.
    if (lstm_batched && prev.used && &prev != &sl)
        UMX_HIP_CHECK(hipStreamWaitEvent(st, prev.ev[ST_COUNT], 0));
    if (int rc = stage_front(sl, st, nb, audio_dev, n, active, nact))
        return rc;
    // two LSTM grids at once only where both fit (the single-track kernel); otherwise wait for the previous
    // segment's last layer
    const bool two_grids = !lstm_batched && 2 * 8 * S <= lstm_capacity;
    for (int layer = 0; layer < 3; ++layer)
    {
        if (layer > 0)
        {
            stage_range(ST_IH0 + 2 * layer);
            UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_IH0 + 2 * layer], st));
            launch_gemm_lanes(sl, st, nb, audio_dev, G_IH, layer, active, nact, false);
        }
        if (prev.used) // the previous segment's layer `layer` must have left its final h/c (F3)
            UMX_HIP_CHECK(hipStreamWaitEvent(st, prev.rec_done[two_grids ? layer : 2], 0));
        stage_range(ST_LSTM0 + 2 * layer);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_LSTM0 + 2 * layer], st));
        if (call_idx < (size_t)kBackupCalls) // the state this layer starts from (the previous segment's layer has finished)
            UMX_HIP_CHECK(hipMemcpyAsync(backup + (call_idx * 3 + layer) * state_floats() * B, state, sizeof(float) * state_floats() * B,
                                         hipMemcpyDeviceToDevice, st));
        if (nact > 0)
            if (int rc = run_lstm_layer(sl, layer, active, nact, flags & UMX_FLAG_LSTM_STEPWISE, lane_mask))
                return rc;
        UMX_HIP_CHECK(hipEventRecord(sl.rec_done[layer], st));
    }
    if (int rc = stage_back(sl, st, nb, audio_dev, out, n, flags, active, nact))
        return rc;
    sl.used = true;
    cur = si;
    ++nseg;
    return UMX_OK;
}
