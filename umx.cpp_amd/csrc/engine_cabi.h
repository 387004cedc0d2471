// engine_cabi.h -- part of engine.hip (one translation unit: the kernels inline into their launchers): the C-ABI of include/umx_hip.h.
// Included by engine.hip behind the definition of umx_hip_ctx; not a stand-alone header.
// ---------------------------------------------------------------- C-ABI
extern "C"
{

int umx_hip_create(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                   const umx_tensor_view *tensors, int n_tensors)
{
    unsigned cf = 0;
    if (const char *e = getenv("UMX_WEIGHTS_RESIDENT")) // lets umx-cli switch without an API change
        if (std::string(e) == "expanded" || std::string(e) == "f32")
            cf |= UMX_CREATE_DEQUANTISE_AT_LOAD;
    if (const char *e = getenv("UMX_GEMM"))
    {
        if (std::string(e) == "f32")
            cf |= UMX_CREATE_GEMM_F32;
        if (std::string(e) == "bf16x3")
            cf |= UMX_CREATE_GEMM_STAGED;
        if (std::string(e) == "planes")
            cf |= UMX_CREATE_GEMM_PLANES;
    }
    if (const char *e = getenv("UMX_LSTM"))
        if (std::string(e) == "batched")
            cf |= UMX_CREATE_LSTM_BATCHED;
    if (const char *e = getenv("UMX_U8"))
        if (std::string(e) == "dequant")
            cf |= UMX_CREATE_U8_DEQUANT;
    return umx_hip_create_ex(out, device, hidden_size, segment_samples, tensors, n_tensors, cf);
}

size_t umx_hip_weight_bytes(const umx_hip_ctx *ctx) { return ctx ? ctx->weight_bytes : 0; }

int umx_hip_create_ex(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                      const umx_tensor_view *tensors, int n_tensors, unsigned create_flags)
{
    return umx_hip_create_tracks(out, device, hidden_size, segment_samples, tensors, n_tensors, create_flags, 1);
}

int umx_hip_create_tracks(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                          const umx_tensor_view *tensors, int n_tensors, unsigned create_flags, int n_tracks)
{
    if (!out || !tensors)
    {
        g_create_error = "umx_hip_create: null argument";
        return UMX_ERR_ARG;
    }
    *out = nullptr;
    umx_hip_ctx *c = new umx_hip_ctx;
    int rc = c->init(device, hidden_size, segment_samples, tensors, n_tensors, create_flags, n_tracks);
    if (rc != UMX_OK)
    {
        g_create_error = c->err;
        umx_hip_destroy(c);
        return rc;
    }
    *out = c;
    return UMX_OK;
}

int umx_hip_n_tracks(const umx_hip_ctx *ctx) { return ctx ? ctx->B : 0; }
int umx_hip_pipeline_depth(const umx_hip_ctx *ctx) { return ctx ? ctx->nslots : 0; }
int umx_hip_lstm_is_batched(const umx_hip_ctx *ctx) { return ctx && ctx->lstm_batched ? 1 : 0; }

unsigned umx_hip_debug_f16_bits(float x) { return f16_rne_bits(x); }

void umx_hip_destroy(umx_hip_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (void *p : ctx->allocs)
        (void)hipFree(p);
    for (hipEvent_t e : ctx->trk_acc_ev)
        if (e)
            (void)hipEventDestroy(e);
    if (ctx->order_ev)
        (void)hipEventDestroy(ctx->order_ev);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    for (int si = 0; si < umx_hip_ctx::kMaxSlots; ++si)
        for (hipEvent_t e : {ctx->slot[si].k_done, ctx->slot[si].out_free})
            if (e)
                (void)hipEventDestroy(e);
    for (int si = 0; si < umx_hip_ctx::kMaxSlots; ++si)
    {
        Slot &sl = ctx->slot[si];
        for (int i = 0; i <= ST_COUNT; ++i)
        {
            if (sl.ev[i])
                (void)hipEventDestroy(sl.ev[i]);
            if (i < ST_COUNT && sl.evk[i])
                (void)hipEventDestroy(sl.evk[i]);
        }
        for (int l = 0; l < 3; ++l)
            if (sl.rec_done[l])
                (void)hipEventDestroy(sl.rec_done[l]);
        if (sl.stream)
            (void)hipStreamDestroy(sl.stream);
    }
    delete ctx;
}

const char *umx_hip_last_error(const umx_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

size_t umx_hip_stream_floats(const umx_hip_ctx *ctx) { return ctx ? ctx->state_floats() : 0; }

// track < 0: every lane
int umx_hip_track_stream_reset(umx_hip_ctx *ctx, int track)
{
    if (!ctx || track >= ctx->B)
        return UMX_ERR_ARG;
    // through umx_hip_sync, not a bare stream wait: a timed-out launch among the calls queued so far is noticed (and
    // repaired by replaying them) BEFORE the state is changed, and the replay log starts afresh behind the change
    if (int rc = umx_hip_sync(ctx))
        return rc;
    const size_t per = ctx->state_floats();
    hipError_t e = track < 0 ? hipMemset(ctx->state, 0, sizeof(float) * per * ctx->B)
                             : hipMemset(ctx->state + per * track, 0, sizeof(float) * per);
    // hipMemset returns before the device has done it, and the slots' non-blocking streams do not order against the null
    // stream: the next segment's kernels must not meet the old state (found in round 3: a rare first-run mismatch)
    if (e == hipSuccess)
        e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    ctx->clear_used(); // nothing in flight: no cross-segment dependency to wait for
    return UMX_OK;
}
int umx_hip_stream_reset(umx_hip_ctx *ctx) { return umx_hip_track_stream_reset(ctx, 0); }

int umx_hip_track_stream_get(umx_hip_ctx *ctx, int track, float *host_dst)
{
    if (!ctx || !host_dst || track < 0 || track >= ctx->B)
        return UMX_ERR_ARG;
    if (int rc = umx_hip_sync(ctx))
        return rc;
    const size_t per = ctx->state_floats();
    hipError_t e = hipMemcpy(host_dst, ctx->state + per * track, sizeof(float) * per, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}
int umx_hip_stream_get(umx_hip_ctx *ctx, float *host_dst) { return umx_hip_track_stream_get(ctx, 0, host_dst); }

int umx_hip_track_stream_set(umx_hip_ctx *ctx, int track, const float *host_src)
{
    if (!ctx || !host_src || track < 0 || track >= ctx->B)
        return UMX_ERR_ARG;
    if (int rc = umx_hip_sync(ctx))
        return rc;
    const size_t per = ctx->state_floats();
    hipError_t e = hipMemcpy(ctx->state + per * track, host_src, sizeof(float) * per, hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    ctx->clear_used();
    return UMX_OK;
}
int umx_hip_stream_set(umx_hip_ctx *ctx, const float *host_src) { return umx_hip_track_stream_set(ctx, 0, host_src); }

size_t umx_hip_stream_layer_floats(const umx_hip_ctx *ctx) { return ctx ? (size_t)4 * 4 * ctx->Hl : 0; }

// one layer's (h, c) of all chains of track lane 0: [target][dir][h|c][Hl]
static int stream_layer_copy(umx_hip_ctx *ctx, int layer, float *host, bool to_host)
{
    if (!ctx || !host || layer < 0 || layer > 2)
        return UMX_ERR_ARG;
    hipError_t e = hipStreamSynchronize(ctx->slot[0].stream);
    if (e == hipSuccess && ctx->ph_next < 0) // outside a phased segment other slots may be busy too
        if (int rc = ctx->sync_all())
            return rc;
    const size_t per = (size_t)4 * ctx->Hl;
    for (int tg = 0; tg < 4 && e == hipSuccess; ++tg)
    {
        float *dev = ctx->state + state_off(tg, layer, 0, 0, ctx->Hl);
        e = to_host ? hipMemcpy(host + tg * per, dev, per * sizeof(float), hipMemcpyDeviceToHost)
                    : hipMemcpy(dev, host + tg * per, per * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}
int umx_hip_stream_get_layer(umx_hip_ctx *ctx, int layer, float *host_dst) { return stream_layer_copy(ctx, layer, host_dst, true); }
int umx_hip_stream_set_layer(umx_hip_ctx *ctx, int layer, const float *host_src)
{
    return stream_layer_copy(ctx, layer, const_cast<float *>(host_src), false);
}

int umx_hip_split_inference(umx_hip_ctx *ctx, const float *audio_host, int length, float *const out_host[4],
                            unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    return ctx ? ctx->track(audio_host, length, -1, out_host, flags, progress, progress_user) : UMX_ERR_ARG;
}
int umx_hip_separate_tracks(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *length, const int *shift_offset,
                            float *const *out_host, unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    return ctx ? ctx->tracks(n_tracks, audio_host, length, shift_offset, out_host, flags, progress, progress_user) : UMX_ERR_ARG;
}
int umx_hip_shift_inference(umx_hip_ctx *ctx, const float *audio_host, int length, int offset, float *const out_host[4],
                            unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    if (!ctx)
        return UMX_ERR_ARG;
    if (offset < 0)
        offset = UMX_REFERENCE_SHIFT; // umx.cpp:115: rand() % 22050, never seeded in the reference (see umx_hip.h)
    return ctx->track(audio_host, length, offset, out_host, flags, progress, progress_user);
}

// testing: the per-bin arithmetic of the Wiener filter -- mix_magnitude, wiener_bin_setup, wiener_bin_apply, the
// functions BOTH filter kernels call -- on caller-given bins, so that a test can hold it against an independent float64
// restatement of wiener.cpp:301-400 in the reference's own operation order (the fused-vs-unfused test compares two callers of
// the same functions).  X [n][2 channels][re, im], masks [n][4 sources][2 channels], R [n][4][R00, Re R01, Im R01, R11],
// y out [n][4][2][re, im].  Needs a current device; no context.
__global__ void debug_wiener_bins_kernel(int n, const float *X, const float *masks, const float *R, float max_abs, float *y)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n)
        return;
    const float2 X0 = make_float2(X[b * 4 + 0], X[b * 4 + 1]), X1 = make_float2(X[b * 4 + 2], X[b * 4 + 3]);
    const float h0 = mix_magnitude(X0), h1 = mix_magnitude(X1);
    float m0[4], m1[4];
    float4 rc[4];
    for (int s = 0; s < 4; ++s)
    {
        m0[s] = masks[(b * 4 + s) * 2 + 0] * h0; // inference.cpp:175-183
        m1[s] = masks[(b * 4 + s) * 2 + 1] * h1;
        rc[s] = make_float4(R[(b * 4 + s) * 4 + 0], R[(b * 4 + s) * 4 + 1], R[(b * 4 + s) * 4 + 2], R[(b * 4 + s) * 4 + 3]);
    }
    WienerBin wb;
    wiener_bin_setup(X0, X1, m0, m1, rc, max_abs, 1.0f / max_abs, wb);
    for (int s = 0; s < 4; ++s)
    {
        float2 o[2];
        wiener_bin_apply(wb, s, rc[s], max_abs, o);
        y[((b * 4 + s) * 2 + 0) * 2 + 0] = o[0].x;
        y[((b * 4 + s) * 2 + 0) * 2 + 1] = o[0].y;
        y[((b * 4 + s) * 2 + 1) * 2 + 0] = o[1].x;
        y[((b * 4 + s) * 2 + 1) * 2 + 1] = o[1].y;
    }
}
int umx_hip_debug_wiener_bins(int n, const float *X, const float *masks, const float *R, float max_abs, float *y)
{
    if (n < 1 || !X || !masks || !R || !y || !(max_abs >= 1.0f))
        return UMX_ERR_ARG;
    float *d = nullptr;
    const size_t nx = (size_t)n * 4, nm = (size_t)n * 8, nr = (size_t)n * 16, ny = (size_t)n * 16;
    if (hipMalloc(reinterpret_cast<void **>(&d), (nx + nm + nr + ny) * sizeof(float)) != hipSuccess)
        return UMX_ERR_HIP;
    bool ok = hipMemcpy(d, X, nx * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d + nx, masks, nm * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d + nx + nm, R, nr * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok)
    {
        hipLaunchKernelGGL(debug_wiener_bins_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, n, d, d + nx, d + nx + nm, max_abs, d + nx + nm + nr);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(y, d + nx + nm + nr, ny * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    return ok ? UMX_OK : UMX_ERR_HIP;
}

// debugging: queue `launches` LDS-guard kernels on a private stream (they run beside whatever the caller
// queues next); read the counters back with launches == 0 (returns words corrupted, events in out2[0..1])
int umx_hip_debug_lds_guard(umx_hip_ctx *ctx, int launches, int rounds, unsigned *out2)
{
    static hipStream_t gs = nullptr;
    static unsigned *errs = nullptr;
    if (!ctx)
        return UMX_ERR_ARG;
    if (!gs)
    {
        if (hipStreamCreateWithFlags(&gs, hipStreamNonBlocking) != hipSuccess || hipMalloc(&errs, 8) != hipSuccess)
            return UMX_ERR_HIP;
        (void)hipMemset(errs, 0, 8);
    }
    for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(lds_guard_kernel, dim3(512), dim3(256), 0, gs, errs, rounds);
    if (launches == 0 && out2)
    {
        (void)hipStreamSynchronize(gs);
        (void)hipMemcpy(out2, errs, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(errs, 0, 8);
    }
    return UMX_OK;
}

int umx_hip_segment_begin(umx_hip_ctx *ctx, const float *audio_host, int n, unsigned flags)
{
    return ctx ? ctx->phase_begin(audio_host, n, flags) : UMX_ERR_ARG;
}
int umx_hip_segment_lstm_layer(umx_hip_ctx *ctx, int layer) { return ctx ? ctx->phase_layer(layer) : UMX_ERR_ARG; }
int umx_hip_segment_end(umx_hip_ctx *ctx, float *const out_host[4]) { return ctx ? ctx->phase_end(out_host) : UMX_ERR_ARG; }

void *umx_hip_phase_stream(umx_hip_ctx *ctx) { return ctx ? (void *)ctx->slot[0].stream : nullptr; }
float *umx_hip_stream_state_device(umx_hip_ctx *ctx) { return ctx ? ctx->state : nullptr; }
int umx_hip_segment_begin_device(umx_hip_ctx *ctx, const float *audio_dev, int n, unsigned flags)
{
    return ctx ? ctx->phase_begin_device(audio_dev, n, flags) : UMX_ERR_ARG;
}
int umx_hip_segment_end_device(umx_hip_ctx *ctx, float *const out_dev[4]) { return ctx ? ctx->phase_end_device(out_dev) : UMX_ERR_ARG; }
int umx_hip_segment_masks_device(umx_hip_ctx *ctx) { return ctx ? ctx->phase_masks() : UMX_ERR_ARG; }
int umx_hip_segment_discard(umx_hip_ctx *ctx)
{
    if (!ctx)
        return UMX_ERR_ARG;
    stage_range(-1);
    ctx->ph_next = -1; // whatever was queued runs to its end; the next segment may begin
    ctx->cur = 0;
    ctx->clear_used();
    return UMX_OK;
}
int umx_hip_segment_finish_device(umx_hip_ctx *ctx, float *const out_dev[4]) { return ctx ? ctx->phase_finish_device(out_dev) : UMX_ERR_ARG; }
float *umx_hip_target_mag_device(umx_hip_ctx *ctx, int target, size_t *floats)
{
    if (!ctx || target < 0 || target > 3)
        return nullptr;
    if (floats)
        *floats = (size_t)2 * ctx->T * MAGP;
    return ctx->slot[0].lane[0].ta[target].mag;
}
int umx_hip_gate_reserve(int device, int cus)
{
    if (device < 0)
        return UMX_ERR_ARG;
    LstmGate &g = g_gate[device & 15];
    std::lock_guard<std::mutex> lock(g.m);
    if (cus > 0)
        g.reservations.push_back(cus);
    else if (cus < 0) // gives ONE request of that size back: another driver's request on the same device stays
    {
        auto it = std::find(g.reservations.begin(), g.reservations.end(), -cus);
        if (it == g.reservations.end())
            return UMX_ERR_ARG;
        g.reservations.erase(it);
    }
    else
        g.reservations.clear();
    // every driver keeps its OWN send / recv kernels resident: what must stay free is the SUM of the outstanding requests
    // (lstm_gate_launch clamps it to half the chip)
    g.reserved = 2 * std::accumulate(g.reservations.begin(), g.reservations.end(), 0);
    return UMX_OK;
}

static Stems4 stems4(float *const p[4])
{
    Stems4 s;
    for (int t = 0; t < 4; ++t)
        s.p[t] = reinterpret_cast<float2 *>(p[t]);
    return s;
}
int umx_hip_weight_stems_device(umx_hip_ctx *ctx, float *const stems_dev[4], int n, void *hip_stream)
{
    if (!ctx || !stems_dev || n < 1)
        return UMX_ERR_ARG;
    hipLaunchKernelGGL(track_weight_kernel, dim3((n + 255) / 256, 4), dim3(256), 0, (hipStream_t)hip_stream, stems4(stems_dev), n, ctx->N);
    return hipGetLastError() == hipSuccess ? UMX_OK : UMX_ERR_HIP;
}
int umx_hip_track_accumulate_device(umx_hip_ctx *ctx, float *const track_dev[4], float *sum_weight_dev,
                                    const float *const weighted_dev[4], int offset, int n, void *hip_stream)
{
    if (!ctx || !track_dev || !sum_weight_dev || !weighted_dev || n < 1 || offset < 0)
        return UMX_ERR_ARG;
    float *w[4] = {const_cast<float *>(weighted_dev[0]), const_cast<float *>(weighted_dev[1]), const_cast<float *>(weighted_dev[2]),
                   const_cast<float *>(weighted_dev[3])};
    hipLaunchKernelGGL(track_add_weighted_kernel, dim3((n + 255) / 256, 4), dim3(256), 0, (hipStream_t)hip_stream, stems4(track_dev),
                       sum_weight_dev, stems4(w), offset, n, ctx->N);
    return hipGetLastError() == hipSuccess ? UMX_OK : UMX_ERR_HIP;
}
int umx_hip_track_normalise_device(umx_hip_ctx *ctx, float *const track_dev[4], const float *sum_weight_dev, int length,
                                   void *hip_stream)
{
    if (!ctx || !track_dev || !sum_weight_dev || length < 1)
        return UMX_ERR_ARG;
    hipLaunchKernelGGL(track_normalise_kernel, dim3((length + 255) / 256, 4), dim3(256), 0, (hipStream_t)hip_stream, stems4(track_dev),
                       sum_weight_dev, 0, length);
    return hipGetLastError() == hipSuccess ? UMX_OK : UMX_ERR_HIP;
}

int umx_hip_infer_segment_device(umx_hip_ctx *ctx, const float *audio_dev, int n, float *const out_dev[4],
                                 unsigned flags)
{
    if (!ctx || !out_dev)
        return UMX_ERR_ARG;
    return ctx->infer_device(audio_dev, n, out_dev, flags);
}

int umx_hip_infer_batch_device(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_dev, const int *n,
                               float *const *out_dev, unsigned flags)
{
    if (!ctx)
        return UMX_ERR_ARG;
    return ctx->infer_batch(n_tracks, audio_dev, n, out_dev, flags);
}

// Ordering against the caller's own HIP streams (the engine alternates between two internal streams)
int umx_hip_order_after(umx_hip_ctx *ctx, void *hip_stream)
{
    if (!ctx)
        return UMX_ERR_ARG;
    hipError_t e = hipSuccess;
    if (!ctx->order_ev)
        e = hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventRecord(ctx->order_ev, (hipStream_t)hip_stream);
    for (int si = 0; si < ctx->nslots && e == hipSuccess; ++si)
        e = hipStreamWaitEvent(ctx->slot[si].stream, ctx->order_ev, 0);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}
int umx_hip_order_before(umx_hip_ctx *ctx, void *hip_stream)
{
    if (!ctx)
        return UMX_ERR_ARG;
    // the caller may recycle the buffers of the calls queued so far once its stream is ordered behind them: they can no
    // longer be replayed, so a timeout among them ends as UMX_ERR_TIMEOUT at the next umx_hip_sync (and the log of
    // queued calls stops growing for callers that never call umx_hip_sync)
    if (!ctx->pending.empty())
    {
        ctx->pending_lost = true;
        ctx->pending.clear();
    }
    hipError_t e = hipSuccess;
    if (!ctx->order_ev)
        e = hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming);
    for (int si = 0; si < ctx->nslots && e == hipSuccess; ++si)
    {
        e = hipEventRecord(ctx->order_ev, ctx->slot[si].stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent((hipStream_t)hip_stream, ctx->order_ev, 0);
    }
    // the host-pointer calls download their stems on a stream of their own: "everything queued so far" includes those copies
    if (ctx->copy_stream && e == hipSuccess)
    {
        e = hipEventRecord(ctx->order_ev, ctx->copy_stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent((hipStream_t)hip_stream, ctx->order_ev, 0);
    }
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}

int umx_hip_sync(umx_hip_ctx *ctx)
{
    if (!ctx)
        return UMX_ERR_ARG;
    if (int rc = ctx->sync_all())
        return rc;
    for (int si = 0; si < ctx->nslots; ++si)
    {
        unsigned *dev_status = ctx->slot[si].status;
        if (!dev_status)
            continue;
        unsigned st = 0;
        hipError_t e = hipMemcpy(&st, dev_status, sizeof st, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
        {
            ctx->set_error(hipGetErrorString(e));
            return UMX_ERR_HIP;
        }
        if (st != 0)
        {
            const std::string what = st == 0x80000000u
                                         ? std::string("persistent LSTM kernel: grid barrier timed out (grid not co-resident)")
                                         : "persistent LSTM kernel timed out waiting for a hidden-state granule (code " + std::to_string(st) + ")";
            for (int sj = 0; sj < ctx->nslots; ++sj)
                (void)hipMemset(ctx->slot[sj].status, 0, sizeof(unsigned));
            (void)hipStreamSynchronize(nullptr); // (null-stream memsets are not ordered against the slots' non-blocking streams)
            ctx->persistent_ok = false; // later launches use the per-step driver
            const size_t ncalls = ctx->pending.size();
            if (!ctx->no_recovery && !ctx->pending_lost && ncalls >= 1 && ncalls <= (size_t)umx_hip_ctx::kBackupCalls)
            {
                const int rc = ctx->recover();
                if (rc == UMX_OK)
                {
                    ctx->set_error("recovered: " + what + "; " + std::to_string(ncalls) + " queued segment call(s) were run again with the "
                                   "per-step LSTM driver (bit-identical), which later calls of this context use as well");
                    return UMX_OK;
                }
            }
            // no way back: the aborted launch left a mix of updated and stale chains behind
            (void)hipMemset(ctx->state, 0, sizeof(float) * ctx->state_floats() * ctx->B);
            (void)hipStreamSynchronize(nullptr);
            ctx->clear_used();
            ctx->pending.clear();
            ctx->pending_lost = false;
            ctx->set_error(what + "; the streaming LSTM state was reset to zero, later segments run the per-step driver");
            return UMX_ERR_TIMEOUT;
        }
    }
    ctx->pending.clear();
    ctx->pending_lost = false;
    return UMX_OK;
}

// Host-pointer forms.  H2D, kernels and D2H are queued on the stream of the pipeline slot the segment runs in, so
// with PINNED host buffers consecutive _async calls overlap one segment's transfers with the other's kernels.
int umx_hip_infer_batch_async(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *n,
                              float *const *out_host, unsigned flags)
{
    if (!ctx || !audio_host || !n || !out_host || n_tracks < 1 || n_tracks > ctx->B)
    {
        if (ctx)
            ctx->set_error("infer: bad arguments");
        return UMX_ERR_ARG;
    }
    const int si = ctx->next_slot();
    if (int rc = ctx->ensure_staging())
        return rc;
    Slot &sl = ctx->slot[si];
    hipStream_t st = sl.stream;
    const float *ain[LSTMB_MAX_TRACKS] = {};
    for (int ln = 0; ln < n_tracks; ++ln)
    {
        if (!audio_host[ln])
            continue;
        if (n[ln] < 1 || n[ln] > ctx->N)
        {
            ctx->set_error("infer_segment: need 1 <= n <= segment_samples");
            return UMX_ERR_ARG;
        }
        float *dst = ctx->stage_in[si] + (size_t)2 * ctx->N * ln;
        hipError_t e = hipMemcpyAsync(dst, audio_host[ln], sizeof(float) * 2 * (size_t)n[ln], hipMemcpyHostToDevice, st);
        if (e != hipSuccess)
        {
            ctx->set_error(hipGetErrorString(e));
            return UMX_ERR_HIP;
        }
        ain[ln] = dst;
    }
    if (int rc = ctx->infer_batch(n_tracks, ain, n, ctx->stage_out[si], flags))
        return rc;
    UMX_HIP_CHECK_CTX(ctx, hipEventRecord(sl.k_done, st));
    if (!ctx->pending_lost && !ctx->pending.empty())
    {
        for (int k = 0; k < 4 * n_tracks; ++k)
            ctx->pending.back().host_out[k] = out_host[k];
        for (int ln = 0; ln < n_tracks; ++ln)
            ctx->pending.back().host_audio[ln] = ain[ln] ? audio_host[ln] : nullptr;
    }
    // the stems go out on the copy stream as soon as they are complete, beside whatever runs next (see queue_download)
    umx_hip_ctx::DeferredDownload d;
    d.valid = true;
    d.si = si;
    d.nb = n_tracks;
    for (int ln = 0; ln < n_tracks; ++ln)
    {
        d.n[ln] = ain[ln] ? n[ln] : 0;
        for (int s = 0; s < 4; ++s)
            d.host[4 * ln + s] = out_host[4 * ln + s];
    }
    return ctx->queue_download(d, ctx->copy_stream);
}

int umx_hip_infer_batch(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *n, float *const *out_host,
                        unsigned flags)
{
    if (int rc = umx_hip_infer_batch_async(ctx, n_tracks, audio_host, n, out_host, flags))
        return rc;
    return umx_hip_sync(ctx);
}

int umx_hip_infer_segment_async(umx_hip_ctx *ctx, const float *audio_host, int n, float *const out_host[4], unsigned flags)
{
    if (!ctx || !audio_host || !out_host)
    {
        if (ctx)
            ctx->set_error("infer_segment: bad arguments");
        return UMX_ERR_ARG;
    }
    return umx_hip_infer_batch_async(ctx, 1, &audio_host, &n, out_host, flags);
}

int umx_hip_infer_segment(umx_hip_ctx *ctx, const float *audio_host, int n, float *const out_host[4], unsigned flags)
{
    if (int rc = umx_hip_infer_segment_async(ctx, audio_host, n, out_host, flags))
        return rc;
    return umx_hip_sync(ctx); // umx_inference returns its outputs: synchronous
}

void *umx_hip_stream_handle(umx_hip_ctx *ctx) { return ctx ? (void *)ctx->slot[ctx->cur].stream : nullptr; }
int umx_hip_nb_frames(const umx_hip_ctx *ctx) { return ctx ? ctx->T : 0; }
int umx_hip_segment_samples(const umx_hip_ctx *ctx) { return ctx ? ctx->N : 0; }
int umx_hip_hidden(const umx_hip_ctx *ctx) { return ctx ? ctx->H : 0; }

long umx_hip_read_tap(umx_hip_ctx *ctx, const char *what, int target, float *dst, size_t cap)
{
    if (!ctx || !what || target < 0 || target > 3)
        return -1;
    std::string w = what;
    const int T = ctx->T, H = ctx->H;
    int which = ctx->cur, lane = 0;
    {
        const size_t hash = w.find('#'); // "name#k": track lane k (default 0)
        if (hash != std::string::npos)
        {
            lane = atoi(w.c_str() + hash + 1);
            const size_t at = w.find('@', hash);
            w = w.substr(0, hash) + (at == std::string::npos ? "" : w.substr(at));
            if (lane < 0 || lane >= ctx->B)
                return -1;
        }
    }
    if (w.size() > 2 && w[w.size() - 2] == '@') // "name@s": pipeline slot s instead of the most recent one
    {
        which = w.back() - '0';
        w = w.substr(0, w.size() - 2);
        if (which < 0 || which >= ctx->nslots)
            return -1;
    }
    const Lane &sl = ctx->slot[which].lane[lane];
    const void *src = nullptr;
    size_t nfl = 0, src_ld = 0, rows = 0, cols = 0; // strided copy when src_ld != cols
    int computed = 0; // 1 |X|, 2 mask x |X|, 3 mask as (T, 4098): formed by a tap kernel into ctx->tap_tmp
    if (w == "spec") { src = sl.spec; nfl = (size_t)2 * 2 * T * NBINS; }
    else if (w == "mix_mag") { src = sl.spec; computed = 1; nfl = (size_t)2 * T * NBINS; }
    else if (w == "x") { src = sl.x; nfl = (size_t)T * KX; }
    else if (w == "fc1") { src = sl.ta[target].cat; rows = T; cols = H; src_ld = 2 * H; nfl = rows * cols; }
    else if (w == "lstm") { src = sl.ta[target].cat + H; rows = T; cols = H; src_ld = 2 * H; nfl = rows * cols; }
    else if (w == "lstm_l0") { src = sl.ta[target].la; nfl = (size_t)T * H; }
    else if (w == "lstm_l1") { src = sl.ta[target].lb; nfl = (size_t)T * H; }
    else if (w == "proj") { src = sl.ta[target].P; nfl = (size_t)T * 4 * H; }
    else if (w == "fc2") { src = sl.ta[target].a2; nfl = (size_t)T * H; }
    else if (w == "mask") { src = sl.ta[target].mag; computed = 3; nfl = (size_t)T * NOUT; }
    else if (w == "target_mag") { src = sl.ta[target].mag; computed = 2; nfl = (size_t)2 * T * NBINS; }
    else if (w == "y") { src = sl.y + (size_t)target * 2 * T * NBINS; nfl = (size_t)2 * 2 * T * NBINS; }
    else if (w == "max_abs") { src = sl.maxabs; nfl = 1; }
    else return -1;
    // the recurrence of a track-batched context writes the fp32 rows of a layer only under UMX_FLAG_DEBUG_TAPS when it also writes
    // the next GEMM's planes: without the flag these taps would be stale rows of an earlier call -- unavailable, like "y"
    if ((w == "lstm" && !ctx->slot[which].lstm_rows_f32[2]) || (w == "lstm_l0" && !ctx->slot[which].lstm_rows_f32[0]) ||
        (w == "lstm_l1" && !ctx->slot[which].lstm_rows_f32[1]))
        return -2;
    if (!src)
        return -2;
    if (!dst)
        return (long)nfl;
    if (cap < nfl)
        return -3;
    if (ctx->sync_all() != UMX_OK)
        return -4;
    if (computed)
    {
        if (!ctx->tap_tmp && ctx->dalloc(&ctx->tap_tmp, (size_t)2 * T * NBINS) != UMX_OK)
            return -4;
        const unsigned blocks = (unsigned)((nfl + 255) / 256);
        if (computed == 1)
            hipLaunchKernelGGL(tap_mix_mag_kernel, dim3(blocks), dim3(256), 0, nullptr, sl.spec, nfl, ctx->tap_tmp);
        else if (computed == 2)
            hipLaunchKernelGGL(tap_target_mag_kernel, dim3(blocks), dim3(256), 0, nullptr, sl.spec, sl.ta[target].mag, nfl, ctx->tap_tmp);
        else
            hipLaunchKernelGGL(tap_mask_kernel, dim3(blocks), dim3(256), 0, nullptr, sl.ta[target].mag, T, ctx->tap_tmp);
        if (hipDeviceSynchronize() != hipSuccess)
            return -4;
        src = ctx->tap_tmp;
    }
    hipError_t e;
    if (rows)
        e = hipMemcpy2D(dst, cols * sizeof(float), src, src_ld * sizeof(float), cols * sizeof(float), rows,
                        hipMemcpyDeviceToHost);
    else
        e = hipMemcpy(dst, src, nfl * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return -4;
    }
    if (w == "max_abs")
    {
        unsigned bits;
        memcpy(&bits, dst, 4);
        float m;
        memcpy(&m, &bits, 4);
        dst[0] = std::max(1.0f, m / WIENER_SCALE);
    }
    return (long)nfl;
}

int umx_hip_stage_times_slot(umx_hip_ctx *ctx, int slot_index, const char **names, float *ms, int cap);
int umx_hip_stage_times(umx_hip_ctx *ctx, const char **names, float *ms, int cap)
{
    return ctx ? umx_hip_stage_times_slot(ctx, ctx->cur, names, ms, cap) : 0;
}

int umx_hip_stage_times_slot(umx_hip_ctx *ctx, int slot_index, const char **names, float *ms, int cap)
{
    if (!ctx || slot_index < 0 || slot_index >= ctx->nslots)
        return 0;
    Slot &sl = ctx->slot[slot_index];
    if (!sl.have_times || ctx->sync_all() != UMX_OK)
        return 0;
    int n = std::min(cap, (int)ST_COUNT);
    for (int i = 0; i < n; ++i)
    {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, sl.ev[i], sl.ev[i + 1]);
        if (names)
            names[i] = kStageNames[i];
        if (ms)
            ms[i] = t;
    }
    return ST_COUNT;
}

// per stage: the time from the event recorded BEHIND the stage's split kernel to the next stage's event, i.e. the stage's main
// kernel alone (GEMM stages of plane contexts); the stage time where there is no such event.  Same conventions as umx_hip_stage_times_slot.
int umx_hip_stage_kernel_times_slot(umx_hip_ctx *ctx, int slot_index, float *ms, int cap)
{
    if (ctx && slot_index < 0)
        slot_index = ctx->cur;
    if (!ctx || slot_index >= ctx->nslots)
        return 0;
    Slot &sl = ctx->slot[slot_index];
    if (!sl.have_times || ctx->sync_all() != UMX_OK)
        return 0;
    const int n = std::min(cap, (int)ST_COUNT);
    for (int i = 0; i < n; ++i)
    {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, sl.evk_set[i] ? sl.evk[i] : sl.ev[i], sl.ev[i + 1]);
        if (ms)
            ms[i] = t;
    }
    return ST_COUNT;
}

const char *umx_hip_lstm_kernel_name(const umx_hip_ctx *ctx) { return ctx ? ctx->lstm_kernel_last : "none"; }

const char *umx_hip_gemm_kernel_name(const umx_hip_ctx *ctx, int mode) { return ctx && mode >= 0 && mode < 4 ? ctx->gemm_kernel_last[mode] : "none"; }

int umx_hip_lstm_was_persistent(const umx_hip_ctx *ctx) { return ctx && ctx->slot[ctx->cur].last_persistent ? 1 : 0; }

int umx_hip_lstm_mode(umx_hip_ctx *ctx)
{
    if (!ctx || !ctx->slot[ctx->cur].last_persistent)
        return 0;
    unsigned st[2] = {0, 0};
    const unsigned *src = ctx->slot[ctx->cur].status;
    if (ctx->sync_all() != UMX_OK || hipMemcpy(st, src, sizeof st, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return st[1] ? 2 : 1;
}

int umx_hip_debug_lstm_profile(umx_hip_ctx *ctx, unsigned long long *out48)
{
    if (!ctx || !out48)
        return UMX_ERR_ARG;
    if (ctx->sync_all() != UMX_OK ||
        hipMemcpy(out48, ctx->slot[ctx->cur].lprof, sizeof(unsigned long long) * 48, hipMemcpyDeviceToHost) != hipSuccess)
        return UMX_ERR_HIP;
    return UMX_OK;
}

// where the workgroups of the last profiled one-track recurrence launch ran: out[i] = xcc << 48 | chain << 40 | slice << 32 | HW_ID
int umx_hip_debug_lstm_placement(umx_hip_ctx *ctx, unsigned long long *out, int n)
{
    if (!ctx || !out || n < 0 || n > 960 + 64 * 8 * 5)
        return UMX_ERR_ARG;
    if (ctx->sync_all() != UMX_OK ||
        hipMemcpy(out, ctx->slot[ctx->cur].lprof + 64, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost) != hipSuccess)
        return UMX_ERR_HIP;
    return UMX_OK;
}

} // extern "C"
