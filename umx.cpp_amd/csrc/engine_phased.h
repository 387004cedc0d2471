// engine_phased.h -- part of engine.hip (one translation unit: the kernels inline into their launchers): the phased segment API (multi-GPU cut points), debug taps, the LDS guard, recovery after a persistent-kernel timeout.
// Included by engine.hip behind the definition of umx_hip_ctx; not a stand-alone header.
// ---------------------------------------------------------------- one segment, phase by phase
// The same launches as infer_device on slot 0, cut where another GPU's LSTM state has to come in: the
// caller sets layer l's incoming (h, c) (umx_hip_stream_set_layer) before phase_layer(l) and reads the
// outgoing one after it.  Used by the exact state-carry pipeline over several GPUs (multigpu.py).
int umx_hip_ctx::phase_begin(const float *audio_host, int n, unsigned flags)
{
    StageRangeCloser close_ranges_on_return;
    if (!audio_host || n < 1 || n > N)
    {
        set_error("segment_begin: need 1 <= n <= segment_samples and non-null audio");
        return UMX_ERR_ARG;
    }
    if (ph_next != -1)
    {
        set_error("segment_begin: a phased segment is already open");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    if (int rc = sync_all())
        return rc;
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(flags, active, nact);
    last_flags = flags;
    UMX_HIP_CHECK(hipMemcpyAsync(audio_in, audio_host, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, sl.stream));
    const float *ain = audio_in;
    if (int rc = stage_front(sl, sl.stream, 1, &ain, &n, active, nact))
        return rc;
    ph_next = 0;
    ph_n = n;
    ph_flags = flags;
    ph_audio = nullptr;
    return UMX_OK;
}

// the same without host transfers and without waiting (multi-GPU driver: everything stays on slot 0's stream)
int umx_hip_ctx::phase_begin_device(const float *audio_dev, int n, unsigned flags)
{
    StageRangeCloser close_ranges_on_return;
    if (!audio_dev || n < 1 || n > N)
    {
        set_error("segment_begin_device: need 1 <= n <= segment_samples and non-null audio");
        return UMX_ERR_ARG;
    }
    if (ph_next != -1)
    {
        set_error("segment_begin: a phased segment is already open");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(flags, active, nact);
    last_flags = flags;
    if (int rc = stage_front(sl, sl.stream, 1, &audio_dev, &n, active, nact))
        return rc;
    ph_next = 0;
    ph_n = n;
    ph_flags = flags;
    ph_audio = audio_dev;
    return UMX_OK;
}

int umx_hip_ctx::phase_end_device(float *const out_dev_[4])
{
    StageRangeCloser close_ranges_on_return;
    if (ph_next != 3 || !out_dev_)
    {
        set_error("segment_end: all three LSTM layers must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(ph_flags, active, nact);
    ph_next = -1;
    const float *ain = ph_audio ? ph_audio : audio_in;
    if (int rc = stage_back(sl, sl.stream, 1, &ain, out_dev_, &ph_n, ph_flags, active, nact))
        return rc;
    cur = 0;
    clear_used();
    return UMX_OK;
}

// fc2 + fc3 of the active targets: their magnitudes are in HBM afterwards (umx_hip_target_mag_device)
int umx_hip_ctx::phase_masks()
{
    StageRangeCloser close_ranges_on_return;
    if (ph_next != 3)
    {
        set_error("segment_masks: all three LSTM layers must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(ph_flags, active, nact);
    const float *ain = ph_audio ? ph_audio : audio_in;
    if (int rc = stage_masks(sl, sl.stream, 1, &ain, ph_flags, active, nact))
        return rc;
    ph_next = 4;
    return UMX_OK;
}

// Wiener + inverse STFT from the magnitudes of all four targets, wherever they came from
int umx_hip_ctx::phase_finish_device(float *const out_dev_[4])
{
    StageRangeCloser close_ranges_on_return;
    if (ph_next != 4 || !out_dev_)
    {
        set_error("segment_finish: umx_hip_segment_masks_device must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    ph_next = -1;
    const float *ain = ph_audio ? ph_audio : audio_in;
    if (int rc = stage_finish(sl, sl.stream, 1, &ain, out_dev_, &ph_n, ph_flags, false))
        return rc;
    cur = 0;
    clear_used();
    return UMX_OK;
}

int umx_hip_ctx::phase_layer(int layer)
{
    StageRangeCloser close_ranges_on_return;
    if (ph_next < 0 || ph_next > 2 || layer != ph_next)
    {
        set_error("segment_lstm_layer: layers run in order 0, 1, 2 after segment_begin");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    hipStream_t st = sl.stream;
    int active[4], nact;
    active_list(ph_flags, active, nact);
    if (layer > 0)
    {
        stage_range(ST_IH0 + 2 * layer);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_IH0 + 2 * layer], st));
        {
            const float *ain = ph_audio ? ph_audio : audio_in;
            launch_gemm_lanes(sl, st, 1, &ain, G_IH, layer, active, nact, false);
        }
    }
    stage_range(ST_LSTM0 + 2 * layer);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_LSTM0 + 2 * layer], st));
    if (nact > 0)
        if (int rc = run_lstm_layer(sl, layer, active, nact, ph_flags & UMX_FLAG_LSTM_STEPWISE, 1u))
            return rc;
    UMX_HIP_CHECK(hipEventRecord(sl.rec_done[layer], st));
    ph_next = layer + 1;
    return UMX_OK;
}

int umx_hip_ctx::phase_end(float *const out_host[4])
{
    StageRangeCloser close_ranges_on_return;
    if (ph_next != 3 || !out_host)
    {
        set_error("segment_end: all three LSTM layers must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(ph_flags, active, nact);
    ph_next = -1;
    const float *ain = audio_in;
    if (int rc = stage_back(sl, sl.stream, 1, &ain, out_dev, &ph_n, ph_flags, active, nact))
        return rc;
    cur = 0;
    if (int rc = umx_hip_sync(this))
        return rc;
    for (int s = 0; s < 4; ++s)
        UMX_HIP_CHECK(hipMemcpy(out_host[s], out_dev[s], sizeof(float) * 2 * (size_t)ph_n, hipMemcpyDeviceToHost));
    clear_used(); // drained: nothing for the next segment to wait for
    return UMX_OK;
}

// ---------------------------------------------------------------- debug taps of what is no longer materialised
// |X| (the STFT kernel keeps only the cropped part the network reads), the target magnitude mask x |X| (formed inside the
// Wiener kernels) and the mask in the reference's (T, 4098) shape: computed on demand, with the SAME device functions the
// hot kernels use (mix_magnitude, common.h), so a tap holds the bits the pipeline works with.
__global__ __launch_bounds__(256) void tap_mix_mag_kernel(const float2 *__restrict__ spec, size_t n, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = mix_magnitude(spec[i]);
}
__global__ __launch_bounds__(256) void tap_target_mag_kernel(const float2 *__restrict__ spec, const float *__restrict__ mask, size_t n,
                                                             float *__restrict__ out) // [2][T][2049]
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = mask[(i / NBINS) * MAGP + i % NBINS] * mix_magnitude(spec[i]); // inference.cpp:175-183
}
__global__ __launch_bounds__(256) void tap_mask_kernel(const float *__restrict__ mask, int T, float *__restrict__ out) // [T][4098]
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * NOUT)
        return;
    const int t = (int)(i / NOUT), k = (int)(i % NOUT), c = k >= NBINS ? 1 : 0;
    out[i] = mask[mask_index(c, T, t, k - c * NBINS)];
}

// ---------------------------------------------------------------- debugging: LDS isolation guard
// A victim workgroup fills 36 KB of LDS with a pattern and keeps verifying it for a while; run beside other
// kernels it shows whether anything else writes into its LDS allocation.
__global__ __launch_bounds__(256) void lds_guard_kernel(unsigned *errs, int rounds)
{
    __shared__ unsigned g[9216]; // 36 KB
    const unsigned salt = blockIdx.x * 2654435761u;
    for (int i = threadIdx.x; i < 9216; i += 256)
        g[i] = salt ^ (unsigned)i;
    __syncthreads();
    for (int r = 0; r < rounds; ++r)
    {
        unsigned bad = 0;
        for (int i = threadIdx.x; i < 9216; i += 256)
            bad += (g[i] != (salt ^ (unsigned)i));
        if (bad)
        {
            atomicAdd(errs, bad);
            atomicAdd(errs + 1, 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 9216; i += 256)
            g[i] = salt ^ (unsigned)i;
        __syncthreads();
        __builtin_amdgcn_s_sleep(32);
    }
}

// After a persistent-kernel timeout (everything has drained): put the stream state back to what it was before the
// first call queued since the last sync, layer by layer, and run those calls again with the per-step driver.
int umx_hip_ctx::recover()
{
    std::vector<PendingCall> calls;
    calls.swap(pending);
    const size_t per = state_floats();
    for (int l = 0; l < 3; ++l) // layer l of every (lane, target): 4 * Hl floats every 12 * Hl
        UMX_HIP_CHECK(hipMemcpy2D(state + (size_t)l * 4 * Hl, sizeof(float) * 12 * Hl, backup + (size_t)l * per * B + (size_t)l * 4 * Hl,
                                  sizeof(float) * 12 * Hl, sizeof(float) * 4 * Hl, (size_t)B * 4, hipMemcpyDeviceToDevice));
    // hipMemset / device-to-device hipMemcpy on the null stream return before the device has done them, and the slots'
    // streams are non-blocking (they do not order against the null stream): without this wait the replayed kernels could
    // read the state before it is restored
    UMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    clear_used();
    recovering = true;
    int rc = UMX_OK;
    for (const PendingCall &pc : calls)
    {
        // host-pointer form: the staging buffer this call read has since been overwritten by the call two later (two
        // staging buffers, up to kBackupCalls calls queued) -- upload the caller's audio again, on the stream the replay of
        // this call is about to be queued on (the earlier user of the buffer ran on the same stream or has been waited for)
        for (int ln = 0; ln < pc.nb; ++ln)
            if (pc.host_audio[ln] && pc.audio[ln])
                UMX_HIP_CHECK(hipMemcpyAsync(const_cast<float *>(pc.audio[ln]), pc.host_audio[ln], sizeof(float) * 2 * (size_t)pc.n[ln],
                                             hipMemcpyHostToDevice, slot[next_slot()].stream));
        if ((rc = infer_batch(pc.nb, pc.audio, pc.n, pc.out, (pc.flags | UMX_FLAG_LSTM_STEPWISE) & ~UMX_FLAG_DEBUG_LSTM_ABORT)) != UMX_OK)
            break;
        for (int k = 0; k < 4 * pc.nb; ++k) // the host-pointer forms had copied the failed run's stems out
            if (pc.host_out[k] && pc.audio[k / 4])
                UMX_HIP_CHECK(hipMemcpyAsync(pc.host_out[k], pc.out[k], sizeof(float) * 2 * (size_t)pc.n[k / 4], hipMemcpyDeviceToHost,
                                             slot[cur].stream));
    }
    recovering = false;
    if (rc == UMX_OK)
        rc = sync_all();
    pending.clear();
    pending_lost = false;
    return rc;
}
