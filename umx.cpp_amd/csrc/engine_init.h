// engine_init.h -- part of engine.hip (one translation unit: the kernels inline into their launchers): weights -> HBM (quantised as stored, exact integer planes, or expanded), every buffer of a context, staging, sync.
// Included by engine.hip behind the definition of umx_hip_ctx; not a stand-alone header.
// ---------------------------------------------------------------- weights
namespace
{
// model.cpp:578-665: q*scale+offset in fp32; F32 passes through
bool dequant(const umx_tensor_view &tv, size_t expect, std::vector<float> &out)
{
    size_t nel = 1;
    for (int i = 0; i < tv.n_dims; ++i)
        nel *= (size_t)tv.ne[i];
    if (nel != expect)
        return false;
    out.resize(nel);
    if (tv.dtype == UMX_DTYPE_F32)
        memcpy(out.data(), tv.data, nel * sizeof(float));
    else if (tv.dtype == UMX_DTYPE_U8)
    {
        const uint8_t *q = static_cast<const uint8_t *>(tv.data);
        for (size_t i = 0; i < nel; ++i)
            out[i] = (float)q[i] * tv.scale + tv.offset;
    }
    else if (tv.dtype == UMX_DTYPE_U16)
    {
        const uint16_t *q = static_cast<const uint16_t *>(tv.data);
        for (size_t i = 0; i < nel; ++i)
            out[i] = (float)q[i] * tv.scale + tv.offset;
    }
    else
        return false;
    return true;
}
} // namespace

int umx_hip_ctx::init(int device_, int hidden, int segment_samples, const umx_tensor_view *tensors, int n_tensors,
                      unsigned create_flags, int n_tracks)
{
    if (n_tracks < 1 || n_tracks > LSTMB_MAX_TRACKS)
    {
        set_error("n_tracks must be in [1, 64]");
        return UMX_ERR_ARG;
    }
    B = n_tracks;
    lstm_batched = B > 1 || (create_flags & UMX_CREATE_LSTM_BATCHED);
    u8_dequant = create_flags & UMX_CREATE_U8_DEQUANT;
    if (hidden <= 0 || hidden % 128 != 0 || hidden > 2048)
    {
        set_error("hidden_size must be a positive multiple of 128 (<= 2048)");
        return UMX_ERR_ARG;
    }
    if (segment_samples < NFFT || segment_samples / HOP + 2 > 4096)
    {
        set_error("segment_samples must be in [4096, 4,190,000] (at most 4094 STFT frames per segment)");
        return UMX_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        set_error("no HIP device available (this engine has no CPU fallback)");
        return UMX_ERR_NODEVICE;
    }
    if (device_ < 0 || device_ >= ndev)
    {
        set_error("device index out of range");
        return UMX_ERR_ARG;
    }
    device = device_;
    UMX_HIP_CHECK(hipSetDevice(device));
    H = hidden;
    Hl = H / 2;
    S = Hl / LSTM_UNITS_PER_WG;
    N = segment_samples;
    T = N / HOP + 1; // dsp.hpp:48
    // rows per track lane of the lane-contiguous activation buffers.  Plane GEMMs run over all lanes at once and their
    // tiles may straddle lanes, so lanes follow each other without padding (T rows; 8 % fewer tile rows than lanes padded
    // to the 256-row tile) and only the END of a launch is padded to a tile (Mpad rows of slack in every buffer);
    // the per-lane GEMM flavours launch M = Tp per lane and need whole 128-row tiles.
    Tp = T; // provisional: fixed below once the GEMM flavour is known
    nbatch = (T + WIENER_BATCH - 1) / WIENER_BATCH;

    // ---- index the tensor views by (target, name)
    std::map<std::string, const umx_tensor_view *> idx[4];
    for (int i = 0; i < n_tensors; ++i)
    {
        if (tensors[i].target < 0 || tensors[i].target > 3 || !tensors[i].name || !tensors[i].data)
        {
            set_error("tensor view with bad target / name / data");
            return UMX_ERR_MODEL;
        }
        idx[tensors[i].target][tensors[i].name] = &tensors[i];
    }
    auto get = [&](int tg, const std::string &name, size_t expect, std::vector<float> &out) -> bool {
        auto it = idx[tg].find(name);
        if (it == idx[tg].end())
        {
            set_error("missing tensor '" + name + "' for target " + std::to_string(tg));
            return false;
        }
        if (!dequant(*it->second, expect, out))
        {
            set_error("tensor '" + name + "' has wrong size in model (target " + std::to_string(tg) + ")");
            return false;
        }
        return true;
    };

    const bool keepq = !(create_flags & UMX_CREATE_DEQUANTISE_AT_LOAD); // u8/u16 views stay as they are (config 5)
    auto view = [&](int tg, const std::string &name) -> const umx_tensor_view * {
        auto it = idx[tg].find(name);
        return it == idx[tg].end() ? nullptr : it->second;
    };
    auto nelems = [](const umx_tensor_view *tv) {
        size_t n = 1;
        for (int i = 0; i < tv->n_dims; ++i)
            n *= (size_t)tv->ne[i];
        return n;
    };
    // Upload `rows` x `cols` of a u8/u16 tensor as stored, into a (rows_pad x cols_pad) device matrix; padding
    // is q = 0 (any finite weight is fine there: padded K columns meet zero activations, padded N rows are
    // never stored).  rowmap (optional) = source row of each destination row.
    auto upload_q = [&](void **dst, const umx_tensor_view *tv, int rows, int cols, int rows_pad, int cols_pad,
                        const std::vector<int> *rowmap, size_t dst_row0, size_t total_rows) -> int {
        const size_t esz = tv->dtype == UMX_DTYPE_U8 ? 1 : 2;
        if (!*dst)
        {
            void *q = nullptr;
            UMX_HIP_CHECK(hipMalloc(&q, total_rows * cols_pad * esz));
            UMX_HIP_CHECK(hipMemset(q, 0, total_rows * cols_pad * esz));
            allocs.push_back(q);
            *dst = q;
            weight_bytes += total_rows * cols_pad * esz;
        }
        std::vector<unsigned char> host((size_t)rows_pad * cols_pad * esz, 0);
        const unsigned char *src = static_cast<const unsigned char *>(tv->data);
        for (int r = 0; r < rows; ++r)
        {
            const int sr = rowmap ? (*rowmap)[r] : r;
            memcpy(&host[(size_t)r * cols_pad * esz], src + (size_t)sr * cols * esz, (size_t)cols * esz);
        }
        UMX_HIP_CHECK(hipMemcpy(static_cast<unsigned char *>(*dst) + dst_row0 * cols_pad * esz, host.data(), host.size(),
                                hipMemcpyHostToDevice));
        return UMX_OK;
    };
    auto is_q = [&](const umx_tensor_view *tv, int dtype, size_t expect) {
        return keepq && tv && tv->dtype == dtype && nelems(tv) == expect;
    };
    // A GEMM launch covers all four targets with ONE kernel instantiation (its B-operand type is a template
    // parameter), so a matrix stays quantised only if it is stored that way for EVERY target; otherwise it is expanded
    // for all of them.
    auto all_q = [&](const std::string &name, int dtype, size_t expect) {
        for (int tg = 0; tg < 4; ++tg)
            if (!is_q(view(tg, name), dtype, expect))
                return false;
        return true;
    };

    if (create_flags & UMX_CREATE_GEMM_F32)
    {
        set_error("UMX_CREATE_GEMM_F32: the fp32-MFMA GEMM flavour was removed in round 3 (slower and, against float64, less accurate "
                  "than the split-operand kernels: profiles/r02_accuracy_vs_float64.txt)");
        return UMX_ERR_ARG;
    }
    gemm_bf16x3 = true; // 16-bit matrix cores with split operands: gemm_planes.h or gemm_bf16x3.h
    // Track-batched contexts fuse the Wiener filter with the inverse STFT (wiener_istft.h: one 1024-thread, 136 KB-LDS
    // workgroup per frame); the single-track context keeps the small kernels, which run beside the other slot's LSTM
    // grids (measured: fused 7.85 ms per segment in the pipeline, unfused 7.41).  UMX_WIENER = fused | stats4 | unfused.
    if (const char *e = getenv("UMX_GEMM_PP"))
        env_gemm_pp = atoi(e);
    if (const char *e = getenv("UMX_GEMM_PS"))
        env_gemm_ps = atoi(e);
    if (const char *e = getenv("UMX_LSTM8_POLL_DELAY")) // tuning: x64 cycles between a wave's publication and its first poll (lstm_batch8.h)
        lstm8_poll_delay = atoi(e);
    if (const char *e = getenv("UMX_LSTM8_PAIRED"))
        env_lstm8_paired = atoi(e) != 0;
    if (const char *e = getenv("UMX_LSTM8_MIN_LANES")) // contexts of this many lanes or more (up to 32) run lstm_batch8_kernel; 99: never
        env_lstm8_min = atoi(e);
    if (const char *e = getenv("UMX_LSTM_POLL_DELAY")) // tuning: x64 cycles a dot wave of the one-track recurrence sleeps before its first poll
        lstm_poll_delay = atoi(e);
    wiener_fused = lstm_batched;
    if (const char *e = getenv("UMX_WIENER")) // fused | stats4 (= statistics kernel + separate filter and inverse-STFT kernels)
        wiener_fused = std::string(e) == "fused";
    // gemm_planes.h for track-batched contexts (large tiles over all lanes); gemm_bf16x3.h for the single-track,
    // latency-optimised context, whose pipeline overlaps small GEMM blocks with two co-resident LSTM grids (the register
    // and LDS budget of DESIGN 4.2 was tuned for exactly that kernel).  Either can be forced.
    gemm_planes = gemm_bf16x3 && ((create_flags & UMX_CREATE_GEMM_PLANES) || (lstm_batched && !(create_flags & UMX_CREATE_GEMM_STAGED)));
    const bool bx = gemm_bf16x3;
    Tp = gemm_planes ? std::max(T, 256) : round_up(T, 128); // (a tile must not hold rows of more than two lanes)
    // A GEMM weight as fp16 planes (PMat): (rows x cols) of `tv` (u8 / u16 as stored: exact integers) or of `f32` (two
    // split terms of w * 2^e, 2^e bringing the tensor's largest |w| into [2^14, 2^15); returns 2^-e), source row
    // rowmap[r] -> destination row dst_row0 + r of a [nbp][total_rows][cols_pad] matrix built in `host`
    auto fill_planes = [&](std::vector<unsigned short> &host, int nbp, size_t total_rows, int cols_pad, const umx_tensor_view *tv,
                           const float *f32, int rows, int cols, const std::vector<int> *rowmap, size_t dst_row0) -> float {
        const size_t plane = total_rows * (size_t)cols_pad;
        if (host.empty())
            host.assign((size_t)nbp * plane, 0);
        float scale = 1.f, unscale = 1.f;
        if (f32)
        {
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)rows * cols; ++i)
                if (std::isfinite(f32[i]))
                    mx = std::max(mx, std::fabs(f32[i]));
            if (mx > 0.f)
            {
                int x;
                (void)std::frexp(mx, &x);
                const int e = std::min(std::max(GP_SPLIT_FIXED_EXP + 1 - x, -100), 100);
                scale = std::ldexp(1.0f, e);
                unscale = std::ldexp(1.0f, -e);
            }
        }
        for (int r = 0; r < rows; ++r)
        {
            const int sr = rowmap ? (*rowmap)[r] : r;
            unsigned short *d = &host[(dst_row0 + r) * cols_pad];
            for (int k = 0; k < cols; ++k)
            {
                if (f32)
                {
                    const float w = f32[(size_t)sr * cols + k] * scale;
                    d[k] = f16_rne_bits(w);
                    d[plane + k] = f16_rne_bits(w - f16_bits_to_float(d[k]));
                }
                else if (nbp == 1)
                    d[k] = f16_rne_bits((float)static_cast<const uint8_t *>(tv->data)[(size_t)sr * cols + k] - 128.0f);
                else
                {
                    // q - 32896 (= 256 (qh - 128) + (ql - 128): the constant of the affine map below) as fp16 + exact remainder:
                    // |remainder| <= 16 = 2^-11 of the plane above it, so that a2 x remainder need not be formed (gemm_planes.h)
                    const float pq = (float)static_cast<const uint16_t *>(tv->data)[(size_t)sr * cols + k] - 32896.0f;
                    d[k] = f16_rne_bits(pq);
                    d[plane + k] = f16_rne_bits(pq - f16_bits_to_float(d[k]));
                }
            }
        }
        return unscale;
    };
    auto upload_pmat = [&](PMat &pm, std::vector<unsigned short> &host, int nbp) -> int {
        pm.nbp = nbp;
        weight_bytes += host.size() * sizeof(unsigned short);
        return upload(&pm.p, host);
    };
    // fp32 matrix (kernel layout, padded) -> device; as three bf16 planes when the bf16x3 GEMMs are selected
    auto upload_matrix = [&](float **dst_f32, unsigned short **dst_bx, const std::vector<float> &w) -> int {
        if (!bx)
        {
            weight_bytes += w.size() * sizeof(float);
            return upload(dst_f32, w);
        }
        std::vector<unsigned short> planes(3 * w.size());
        for (size_t i = 0; i < w.size(); ++i)
            split3_host(w[i], planes[i], planes[w.size() + i], planes[2 * w.size() + i]);
        weight_bytes += planes.size() * sizeof(unsigned short);
        return upload(dst_bx, planes);
    };

    const int G = 4 * Hl; // gate rows per direction
    std::vector<float> whh_h[3], bhh_h[3];
    std::vector<unsigned char> whh_qh[3];
    bool whh_all_u8 = keepq;
    for (int tg = 0; tg < 4 && whh_all_u8; ++tg)
        for (int l = 0; l < 3; ++l)
            for (int dir = 0; dir < 2; ++dir)
            {
                const umx_tensor_view *tv = view(tg, "lstm.weight_hh_l" + std::to_string(l) + (dir ? "_reverse" : ""));
                whh_all_u8 = whh_all_u8 && tv && tv->dtype == UMX_DTYPE_U8 && nelems(tv) == (size_t)G * Hl;
            }
    if (whh_all_u8)
        for (int l = 0; l < 3; ++l)
            whh_qh[l].assign((size_t)8 * S * Hl * 64, 0);
    for (int l = 0; l < 3; ++l)
    {
        whh_h[l].assign((size_t)8 * S * Hl * 64, 0.f);
        bhh_h[l].assign((size_t)8 * S * 64, 0.f);
    }
    for (int tg = 0; tg < 4; ++tg)
    {
        TargetBufs &b = tb[tg];
        std::vector<float> v, w;
        // input / output scaling, duplicated per channel like model.cpp:240-290
        if (!get(tg, "input_scale", CROP, v))
            return UMX_ERR_MODEL;
        w.assign(KX, 0.f);
        for (int k = 0; k < NIN; ++k)
            w[k] = v[k % CROP];
        if (int rc = upload(&b.in_scale, w))
            return rc;
        if (!get(tg, "input_mean", CROP, v))
            return UMX_ERR_MODEL;
        w.assign(KX, 0.f);
        for (int k = 0; k < NIN; ++k)
            w[k] = v[k % CROP];
        if (int rc = upload(&b.in_mean, w))
            return rc;
        if (!get(tg, "output_scale", NBINS, v))
            return UMX_ERR_MODEL;
        // fc3's columns: channel c at [c * MAGP, c * MAGP + 2049) (gemm_common.h); model.cpp:240-290 duplicates per channel
        w.assign(NOUT_PAD, 0.f);
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < NBINS; ++k)
                w[c * MAGP + k] = v[k];
        if (int rc = upload(&b.out_scale, w))
            return rc;
        if (!get(tg, "output_mean", NBINS, v))
            return UMX_ERR_MODEL;
        w.assign(NOUT_PAD, 0.f);
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < NBINS; ++k)
                w[c * MAGP + k] = v[k];
        if (int rc = upload(&b.out_mean, w))
            return rc;
        // fc1 (H x 2974) -> (H x KX), zero K padding
        const bool exact_ok = keepq && !(create_flags & UMX_CREATE_U8_DEQUANT); // integers as exact bf16 planes
        if (gemm_planes)
        {
            std::vector<unsigned short> host;
            const umx_tensor_view *tv = view(tg, "fc1.weight");
            if (exact_ok && all_q("fc1.weight", UMX_DTYPE_U8, (size_t)H * NIN))
            {
                fill_planes(host, 1, H, KX, tv, nullptr, H, NIN, nullptr, 0);
                b.fc1_p.s[0] = tv->scale;
                b.fc1_p.o2[0] = tv->offset + 128.0f * tv->scale;
                if (int rc = upload_pmat(b.fc1_p, host, 1))
                    return rc;
            }
            else
            {
                if (!get(tg, "fc1.weight", (size_t)H * NIN, v))
                    return UMX_ERR_MODEL;
                b.fc1_p.s[0] = fill_planes(host, 2, H, KX, nullptr, v.data(), H, NIN, nullptr, 0);
                if (int rc = upload_pmat(b.fc1_p, host, 2))
                    return rc;
            }
        }
        else if (const umx_tensor_view *tv = view(tg, "fc1.weight"); all_q("fc1.weight", UMX_DTYPE_U8, (size_t)H * NIN))
        {
            if (int rc = upload_q(&b.fc1_q.q, tv, H, NIN, H, KX, nullptr, 0, H))
                return rc;
            b.fc1_q.type = BQ_U8;
            b.fc1_q.s[0] = tv->scale;
            b.fc1_q.o[0] = tv->offset;
        }
        else
        {
            if (!get(tg, "fc1.weight", (size_t)H * NIN, v))
                return UMX_ERR_MODEL;
            w.assign((size_t)H * KX, 0.f);
            for (int o = 0; o < H; ++o)
                memcpy(&w[(size_t)o * KX], &v[(size_t)o * NIN], sizeof(float) * NIN);
            if (int rc = upload_matrix(&b.fc1_w, &b.fc1_bx, w))
                return rc;
        }
        const char *bnn[4] = {"running_mean", "running_var", "weight", "bias"};
        for (int k = 0; k < 4; ++k)
        {
            if (!get(tg, std::string("bn1.") + bnn[k], H, v))
                return UMX_ERR_MODEL;
            if (int rc = upload(&b.bn1[k], v))
                return rc;
            if (!get(tg, std::string("bn2.") + bnn[k], H, v))
                return UMX_ERR_MODEL;
            if (int rc = upload(&b.bn2[k], v))
                return rc;
            if (!get(tg, std::string("bn3.") + bnn[k], NOUT, v))
                return UMX_ERR_MODEL;
            w.assign(NOUT_PAD, k == 1 ? 1.f : 0.f); // padded running_var = 1: no 0/0 in dead columns
            for (int c = 0; c < 2; ++c)
                memcpy(&w[c * MAGP], &v[c * NBINS], sizeof(float) * NBINS);
            if (int rc = upload(&b.bn3[k], w))
                return rc;
        }
        if (gemm_planes)
        {
            std::vector<unsigned short> host;
            const umx_tensor_view *tv = view(tg, "fc2.weight");
            if (exact_ok && all_q("fc2.weight", UMX_DTYPE_U16, (size_t)H * 2 * H))
            {
                fill_planes(host, 2, H, 2 * H, tv, nullptr, H, 2 * H, nullptr, 0);
                b.fc2_p.s[0] = tv->scale;
                b.fc2_p.o2[0] = tv->offset + 32896.0f * tv->scale;
                if (int rc = upload_pmat(b.fc2_p, host, 2))
                    return rc;
            }
            else
            {
                if (!get(tg, "fc2.weight", (size_t)H * 2 * H, v))
                    return UMX_ERR_MODEL;
                b.fc2_p.s[0] = fill_planes(host, 2, H, 2 * H, nullptr, v.data(), H, 2 * H, nullptr, 0);
                if (int rc = upload_pmat(b.fc2_p, host, 2))
                    return rc;
            }
        }
        else if (const umx_tensor_view *tv = view(tg, "fc2.weight"); all_q("fc2.weight", UMX_DTYPE_U16, (size_t)H * 2 * H))
        {
            if (int rc = upload_q(&b.fc2_q.q, tv, H, 2 * H, H, 2 * H, nullptr, 0, H))
                return rc;
            b.fc2_q.type = BQ_U16;
            b.fc2_q.s[0] = tv->scale;
            b.fc2_q.o[0] = tv->offset;
        }
        else
        {
            if (!get(tg, "fc2.weight", (size_t)H * 2 * H, v))
                return UMX_ERR_MODEL;
            if (int rc = upload_matrix(&b.fc2_w, &b.fc2_bx, v))
                return rc;
        }
        // fc3's output rows in the column layout of the mask planes: channel c's 2049 rows at [c * MAGP, ...), zero rows between
        std::vector<unsigned char> fc3_perm;
        umx_tensor_view fc3_tv;
        memset(&fc3_tv, 0, sizeof fc3_tv);
        if (const umx_tensor_view *tv = view(tg, "fc3.weight"); tv && tv->dtype == UMX_DTYPE_U16 && nelems(tv) == (size_t)NOUT * H)
        {
            fc3_perm.assign((size_t)NOUT_PAD * H * 2, 0);
            for (int c = 0; c < 2; ++c)
                memcpy(&fc3_perm[(size_t)c * MAGP * H * 2], static_cast<const unsigned char *>(tv->data) + (size_t)c * NBINS * H * 2, (size_t)NBINS * H * 2);
            fc3_tv = *tv;
            fc3_tv.data = fc3_perm.data();
        }
        auto fc3_f32 = [&](std::vector<float> &dst) -> bool { // dequantised fp32, permuted, (NOUT_PAD x H)
            std::vector<float> src;
            if (!get(tg, "fc3.weight", (size_t)NOUT * H, src))
                return false;
            dst.assign((size_t)NOUT_PAD * H, 0.f);
            for (int c = 0; c < 2; ++c)
                memcpy(&dst[(size_t)c * MAGP * H], &src[(size_t)c * NBINS * H], sizeof(float) * (size_t)NBINS * H);
            return true;
        };
        if (gemm_planes)
        {
            std::vector<unsigned short> host;
            const umx_tensor_view *tv = fc3_tv.data ? &fc3_tv : nullptr;
            if (exact_ok && all_q("fc3.weight", UMX_DTYPE_U16, (size_t)NOUT * H))
            {
                fill_planes(host, 2, NOUT_PAD, H, tv, nullptr, NOUT_PAD, H, nullptr, 0);
                b.fc3_p.s[0] = tv->scale;
                b.fc3_p.o2[0] = tv->offset + 32896.0f * tv->scale;
                if (int rc = upload_pmat(b.fc3_p, host, 2))
                    return rc;
            }
            else
            {
                if (!fc3_f32(v))
                    return UMX_ERR_MODEL;
                b.fc3_p.s[0] = fill_planes(host, 2, NOUT_PAD, H, nullptr, v.data(), NOUT_PAD, H, nullptr, 0);
                if (int rc = upload_pmat(b.fc3_p, host, 2))
                    return rc;
            }
        }
        else if (const umx_tensor_view *tv = &fc3_tv; all_q("fc3.weight", UMX_DTYPE_U16, (size_t)NOUT * H))
        {
            if (int rc = upload_q(&b.fc3_q.q, tv, NOUT_PAD, H, NOUT_PAD, H, nullptr, 0, NOUT_PAD))
                return rc;
            b.fc3_q.type = BQ_U16;
            b.fc3_q.s[0] = tv->scale;
            b.fc3_q.o[0] = tv->offset;
        }
        else
        {
            if (!fc3_f32(w))
                return UMX_ERR_MODEL;
            if (int rc = upload_matrix(&b.fc3_w, &b.fc3_bx, w))
                return rc;
        }
        // LSTM: permute gate rows so a workgroup's 64 columns (g,u) are contiguous
        for (int l = 0; l < 3; ++l)
        {
            const umx_tensor_view *ihv[2] = {view(tg, "lstm.weight_ih_l" + std::to_string(l)),
                                             view(tg, "lstm.weight_ih_l" + std::to_string(l) + "_reverse")};
            const bool ih_q8 = all_q("lstm.weight_ih_l" + std::to_string(l), UMX_DTYPE_U8, (size_t)G * H) &&
                               all_q("lstm.weight_ih_l" + std::to_string(l) + "_reverse", UMX_DTYPE_U8, (size_t)G * H);
            const bool ih_exact = gemm_planes && exact_ok && ih_q8;
            const bool ih_q = ih_q8 && (!gemm_planes || ih_exact); // the source stays u8 (no fp32 copy needed)
            std::vector<unsigned short> ih_planes;
            std::vector<float> ihw((ih_q || gemm_planes) ? 0 : (size_t)2 * G * H), ihb((size_t)2 * G);
            for (int dir = 0; dir < 2; ++dir)
            {
                const std::string sfx = "_l" + std::to_string(l) + (dir ? "_reverse" : "");
                std::vector<float> wih, whhv, bih, bhhv;
                if ((!ih_q && !get(tg, "lstm.weight_ih" + sfx, (size_t)G * H, wih)) ||
                    (!whh_all_u8 && !get(tg, "lstm.weight_hh" + sfx, (size_t)G * Hl, whhv)) ||
                    !get(tg, "lstm.bias_ih" + sfx, G, bih) || !get(tg, "lstm.bias_hh" + sfx, G, bhhv))
                    return UMX_ERR_MODEL;
                const int chain = tg * 2 + dir;
                const umx_tensor_view *hhv = view(tg, "lstm.weight_hh" + sfx);
                const unsigned char *hhq = whh_all_u8 ? static_cast<const unsigned char *>(hhv->data) : nullptr;
                if (whh_all_u8)
                {
                    whh_s[l][chain] = hhv->scale;
                    whh_o[l][chain] = hhv->offset;
                }
                std::vector<int> rowmap(G); // destination gate-interleaved row -> PyTorch gate row
                for (int sl = 0; sl < S; ++sl)
                    for (int g = 0; g < 4; ++g)
                        for (int u = 0; u < 16; ++u)
                        {
                            const int row = g * Hl + sl * 16 + u; // PyTorch gate row (i|f|g|o blocks)
                            const int col = u * 4 + g; // the 4 gates of a unit share a DPP quad
                            const size_t n = (size_t)dir * G + (size_t)sl * 64 + col;
                            rowmap[sl * 64 + col] = row;
                            if (!ih_q && !gemm_planes)
                                memcpy(&ihw[n * H], &wih[(size_t)row * H], sizeof(float) * H);
                            ihb[n] = bih[row];
                            bhh_h[l][((size_t)chain * S + sl) * 64 + col] = bhhv[row];
                            for (int k = 0; k < Hl; ++k)
                            {
                                const size_t di = (((size_t)chain * S + sl) * Hl + k) * 64 + col;
                                if (whh_all_u8)
                                    whh_qh[l][di] = hhq[(size_t)row * Hl + k];
                                else
                                    whh_h[l][di] = whhv[(size_t)row * Hl + k];
                            }
                        }
                if (gemm_planes)
                {
                    if (ih_exact)
                    {
                        fill_planes(ih_planes, 1, (size_t)2 * G, H, ihv[dir], nullptr, G, H, &rowmap, (size_t)dir * G);
                        b.ih_p[l].s[dir] = ihv[dir]->scale;
                        b.ih_p[l].o2[dir] = ihv[dir]->offset + 128.0f * ihv[dir]->scale;
                    }
                    else
                        b.ih_p[l].s[dir] = fill_planes(ih_planes, 2, (size_t)2 * G, H, nullptr, wih.data(), G, H, &rowmap, (size_t)dir * G);
                }
                else if (ih_q)
                {
                    if (int rc = upload_q(&b.ih_q[l].q, ihv[dir], G, H, G, H, &rowmap, (size_t)dir * G, (size_t)2 * G))
                        return rc;
                    b.ih_q[l].type = BQ_U8;
                    b.ih_q[l].s[dir] = ihv[dir]->scale;
                    b.ih_q[l].o[dir] = ihv[dir]->offset;
                }
            }
            if (gemm_planes)
            {
                if (int rc = upload_pmat(b.ih_p[l], ih_planes, ih_exact ? 1 : 2))
                    return rc;
            }
            else if (!ih_q)
            {
                if (int rc = upload_matrix(&b.ih_w[l], &b.ih_bx[l], ihw))
                    return rc;
            }
            if (int rc = upload(&b.ih_b[l], ihb))
                return rc;
        }
    }
    for (int l = 0; l < 3; ++l)
    {
        if (whh_all_u8)
        {
            if (int rc = upload(&whh_q[l], whh_qh[l]))
                return rc;
            weight_bytes += whh_qh[l].size();
        }
        else
        {
            if (int rc = upload(&whh[l], whh_h[l]))
                return rc;
            weight_bytes += whh_h[l].size() * sizeof(float);
        }
        if (int rc = upload(&bhh[l], bhh_h[l]))
            return rc;
    }
    // ---- tables: window (dsp.hpp:61-78, the reference's float PI), window sum-square
    // (dsp.hpp:80-101, same accumulation order), FFT twiddles (rounded from double)
    {
        std::vector<float> w(NFFT);
        static const float PI = 3.14159265359F;
        const float floatN = (float)(NFFT + 1);
        for (int n = 0; n < NFFT; ++n)
            w[n] = 0.5F * (1.0F - cosf(2.0F * PI * (float)n / (floatN - 1)));
        if (int rc = upload(&window, w))
            return rc;
        const size_t total = (size_t)NFFT + (size_t)HOP * (T - 1);
        std::vector<float> nwh(total, 0.f);
        for (int i = 0; i < T; ++i)
        {
            const size_t s0 = (size_t)i * HOP;
            for (size_t j = s0; j < std::min(total, s0 + NFFT); ++j)
                nwh[j] += w[j - s0] * w[j - s0];
        }
        if (int rc = upload(&nw, nwh))
            return rc;
        std::vector<float2> t1(256), t2(4096);
        for (int r = 0; r < 16; ++r)
            for (int k = 0; k < 16; ++k)
            {
                const double ph = -2.0 * M_PI * (double)(r * k) / 256.0;
                t1[r * 16 + k] = make_float2((float)cos(ph), (float)sin(ph));
            }
        for (int r = 0; r < 16; ++r)
            for (int j = 0; j < 256; ++j)
            {
                const double ph = -2.0 * M_PI * (double)(r * j) / 4096.0;
                t2[r * 256 + j] = make_float2((float)cos(ph), (float)sin(ph));
            }
        if (int rc = upload(&tw1, t1))
            return rc;
        if (int rc = upload(&tw2, t2))
            return rc;
    }
    // ---- per-segment buffers: two pipeline slots x B track lanes
    if (int rc = dalloc(&audio_in, (size_t)2 * N))
        return rc;
    for (int k = 0; k < 4; ++k)
        if (int rc = dalloc(&out_dev[k], (size_t)2 * N))
            return rc;
    if (int rc = dalloc(&state, state_floats() * B))
        return rc;
    if (int rc = dalloc(&state_alt, state_floats() * B))
        return rc;
    if (int rc = dalloc(&backup, (size_t)kBackupCalls * 3 * state_floats() * B))
        return rc;
    lsync_words = LSTM_SYNC_HEADER_WORDS + std::max(granule_count(S) * 2, lstm_batched ? lstmb_granule_words(Hl) * ((B + LSTMB_GROUP_TRACKS - 1) / LSTMB_GROUP_TRACKS) : (size_t)0);
    // Two slots.  (Three were tried for single-track contexts in round 3 -- a third segment in flight has its front stage
    // done by the time an LSTM grid retires, so that two grids would be resident all the time: 7.57 ms per segment against
    // 6.70 with two; the grids and the GEMM blocks beside them only slow each other down, avg LSTM launch 3.03 -> 3.79 ms.)
    nslots = 2;
    if (gemm_planes)
    {
        // the plane GEMMs address their operands and fc3's mask output through buffer resources with 32-bit byte offsets
        // (gemm_planes.h: both planes of an operand behind one base; gemm_common.h: all lanes' masks behind one base): refuse
        // lane x segment-length combinations that do not fit instead of reading zeros past the range check
        const unsigned long long rows = (unsigned long long)B * Tp + Mpad;
        const unsigned long long planes_bytes = 2ull * rows * (unsigned long long)std::max(KX, 2 * H) * 2ull;
        const unsigned long long mask_bytes = (unsigned long long)B * 2ull * T * MAGP * 4ull;
        if (planes_bytes >= (1ull << 31) || mask_bytes >= (1ull << 32))
        {
            set_error("track lanes x segment length exceed the 32-bit addressing of the plane GEMMs' operands: fewer lanes or a shorter segment");
            return UMX_ERR_ARG;
        }
    }
    for (int si = 0; si < nslots; ++si)
    {
        Slot &sl = slot[si];
        UMX_HIP_CHECK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        // Buffers that a launch covering several track lanes reads or writes (the batched LSTM kernel, the plane GEMMs
        // with M = lanes x Tp) are ONE allocation per slot (and target), lane after lane at a constant stride.
        float *x_all = nullptr;
        if (int rc = dalloc(&x_all, ((size_t)B * Tp + Mpad) * KX))
            return rc;
        for (int tg = 0; tg < 4; ++tg)
        {
            float *cat_all = nullptr, *la_all = nullptr, *lb_all = nullptr, *P_all = nullptr, *a2_all = nullptr, *mag_all = nullptr;
            if (int rc = dalloc(&cat_all, ((size_t)B * Tp + Mpad) * 2 * H))
                return rc;
            if (int rc = dalloc(&la_all, ((size_t)B * Tp + Mpad) * H))
                return rc;
            if (int rc = dalloc(&lb_all, ((size_t)B * Tp + Mpad) * H))
                return rc;
            if (int rc = dalloc(&P_all, ((size_t)B * Tp + Mpad) * 4 * H))
                return rc;
            if (int rc = dalloc(&a2_all, ((size_t)B * Tp + Mpad) * H))
                return rc;
            if (int rc = dalloc(&mag_all, (size_t)B * 2 * T * MAGP))
                return rc;
            unsigned short *xs_p = nullptr, *cat_p = nullptr, *la_p = nullptr, *lb_p = nullptr, *a2_p = nullptr;
            float *rs[8] = {};
            if (gemm_planes) // planes [2][B * Tp][K]: plane-major over ALL lanes, so that M runs across the lanes
            {
                if (int rc = dalloc(&xs_p, 2 * ((size_t)B * Tp + Mpad) * KX))
                    return rc;
                if (int rc = dalloc(&cat_p, 2 * ((size_t)B * Tp + Mpad) * 2 * H))
                    return rc;
                for (unsigned short **q : {&la_p, &lb_p, &a2_p})
                    if (int rc = dalloc(q, 2 * ((size_t)B * Tp + Mpad) * H))
                        return rc;
                for (int k = 0; k < 8; ++k) // [2..4]: the recurrence's outputs, one array per direction when it writes the planes itself
                    if (int rc = dalloc(&rs[k], ((size_t)B * Tp + Mpad) * (k >= 2 && k <= 4 ? 2 : 1)))
                        return rc;
            }
            for (int ln = 0; ln < B; ++ln)
            {
                TargetAct &b = sl.lane[ln].ta[tg];
                b.cat = cat_all + (size_t)ln * Tp * 2 * H;
                b.la = la_all + (size_t)ln * Tp * H;
                b.lb = lb_all + (size_t)ln * Tp * H;
                b.P = P_all + (size_t)ln * Tp * 4 * H;
                b.a2 = a2_all + (size_t)ln * Tp * H;
                b.mag = mag_all + (size_t)ln * 2 * T * MAGP;
                if (gemm_planes) // lane ln's rows start at row ln * Tp of every plane
                {
                    b.xs_p = xs_p + (size_t)ln * Tp * KX;
                    b.cat_p = cat_p + (size_t)ln * Tp * 2 * H;
                    b.la_p = la_p + (size_t)ln * Tp * H;
                    b.lb_p = lb_p + (size_t)ln * Tp * H;
                    b.a2_p = a2_p + (size_t)ln * Tp * H;
                    b.rs_xs = rs[0] + (size_t)ln * Tp;
                    b.rs_catL = rs[1] + (size_t)ln * Tp;
                    b.rs_catR = rs[2] + (size_t)ln * Tp;
                    b.rs_la = rs[3] + (size_t)ln * Tp;
                    b.rs_lb = rs[4] + (size_t)ln * Tp;
                    b.rs_a2 = rs[5] + (size_t)ln * Tp;
                    b.rsc_xs = rs[6] + (size_t)ln * Tp;
                    b.rsc_a2 = rs[7] + (size_t)ln * Tp;
                }
            }
        }
        {
            // every lane's streaming buffers a fixed stride apart (lane_strides()): the streaming kernels take lane 0's
            // pointers and cover all active lanes in one launch (common.h LaneSet)
            const WienerStrides ls = lane_strides();
            float2 *spec_all, *y_all, *frames_all;
            float *wpart_all, *Rc_all;
            unsigned *maxabs_all;
            if (int rc = dalloc(&spec_all, (size_t)B * ls.spec))
                return rc;
            if (int rc = dalloc(&y_all, (size_t)B * ls.y))
                return rc;
            if (int rc = dalloc(&frames_all, (size_t)B * ls.frames))
                return rc;
            if (int rc = dalloc(&wpart_all, (size_t)B * ls.part))
                return rc;
            if (int rc = dalloc(&Rc_all, (size_t)B * ls.rc))
                return rc;
            if (int rc = dalloc(&maxabs_all, (size_t)B))
                return rc;
            for (int ln = 0; ln < B; ++ln)
            {
                Lane &L = sl.lane[ln];
                L.x = x_all + (size_t)ln * Tp * KX;
                L.spec = spec_all + (size_t)ln * ls.spec;
                L.y = y_all + (size_t)ln * ls.y;
                L.frames = frames_all + (size_t)ln * ls.frames;
                L.wpart = wpart_all + (size_t)ln * ls.part;
                L.Rc = Rc_all + (size_t)ln * ls.rc;
                L.maxabs = maxabs_all + ln;
            }
        }
        if (int rc = dalloc(&sl.status, 4))
            return rc;
        if (int rc = dalloc(&sl.hbuf, (size_t)2 * 8 * Hl))
            return rc;
        if (int rc = dalloc(&sl.lsync, lsync_words))
            return rc;
        if (int rc = dalloc(&sl.lprof, 1024 + 64 * 8 * 5))
            return rc;
        for (int i = 0; i <= ST_COUNT; ++i)
            UMX_HIP_CHECK(hipEventCreate(&sl.ev[i]));
        for (int i : {ST_FC1, ST_IH0, ST_IH1, ST_IH2, ST_FC2, ST_FC3})
            UMX_HIP_CHECK(hipEventCreate(&sl.evk[i]));
        for (int l = 0; l < 3; ++l)
            UMX_HIP_CHECK(hipEventCreateWithFlags(&sl.rec_done[l], hipEventDisableTiming));
    }
    stream = slot[0].stream;
    {
        // residency of the persistent LSTM kernel: the smaller of the two activation flavours of the
        // instantiation this hidden size uses (the two-grid co-residency of the pipeline rests on it)
        const int kpw = Hl / 8;
        int per_cu = 1 << 30, cus = 0;
        for (int precise = 0; precise < 2; ++precise)
        {
            const void *fn = lstm_persistent_fn(kpw, precise != 0);
            int v = 0;
            UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, lstm_threads, 0));
            per_cu = std::min(per_cu, v);
        }
        UMX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        n_cus = cus;
        lstm_capacity = per_cu * cus;
        if (lstm_batched)
        {
            // the batched kernel: worst-case dynamic LDS (16 lanes), both activation flavours
            const size_t lds_max = lstmb_lds_bytes(LSTMB_GROUP_TRACKS, 8);
            per_cu = 1 << 30;
            for (int precise = 0; precise < 2; ++precise)
                for (int wq = 0; wq < 2; ++wq)
                {
                    const void *fn = lstm_batch_fn(Hl, wq != 0, precise != 0);
                    if (!fn)
                    {
                        set_error("track batching needs hidden_size in {128, 256, 512, 1024}");
                        return UMX_ERR_ARG;
                    }
                    UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
                    int v = 0;
                    UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, LSTM_THREADS, lstmb_lds_bytes(B > 8 ? 16 : B > 4 ? 8 : B > 2 ? 4 : B > 1 ? 2 : 1, B > 8 ? 8 : 16)));
                    per_cu = std::min(per_cu, v);
                }
            // octets of 8 lanes x column shards of 64 units (lstm_batch8.h): LSTM hidden 512 / 256 with the u8-resident W_hh, chosen per
            // CONTEXT and used for every launch of it -- its sums are not the bits of lstm_batch_kernel, and a lane's result must not depend on
            // who rides along.  Everything else (hidden 256 / 128, fp32-resident W_hh, UMX_LSTM8_MIN_LANES=99) runs lstm_batch_kernel, a group
            // of 16 lanes per launch, the groups one after the other.
            lstm_batch8_ok = false;
            if (B >= env_lstm8_min && lstm_batch8_fn(Hl, false) && !u8_dequant && whh_q[0] && whh_q[1] && whh_q[2])
            {
                lstm_batch8_ok = true;
                int per_cu8 = 1 << 30;
                for (int precise = 0; precise < 2; ++precise)
                    for (int no = 1; no <= 2; ++no)
                    {
                        const void *fn = lstm_batch8_fn(Hl, precise != 0, no);
                        UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lstm8_lds_bytes(Hl, no)));
                        int v = 0;
                        UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, LSTM_THREADS, lstm8_lds_bytes(Hl, no)));
                        per_cu8 = std::min(per_cu8, v);
                    }
                lstm_batch8_ok = per_cu8 >= 1 && 8 * 32 <= per_cu8 * cus;
                if (lstm_batch8_ok) // the persistent launches of this context are lstm_batch8_kernel's: its occupancy decides
                    per_cu = per_cu8;
            }
            lstm_batch_capacity = per_cu * cus;
            // the recurrence writes the plane GEMMs' A operands (layers 1, 2 and fc2's right half) and their row sums itself where
            // it runs on the u8-resident W_hh (its gate lanes hold h as two fp16 planes, its all-ones tile the row sums): no
            // split_planes launches for them (lstm_batch.h, LstmBArgs::planes).  -DUMX_FUSE_LSTM_PLANES=0: A/B builds
            lstm_rowsums = UMX_FUSE_LSTM_PLANES && gemm_planes && !u8_dequant && whh_q[0] && whh_q[1] && whh_q[2];
            lstm_writes_planes = lstm_rowsums;
        }
    }
    // dynamic LDS > 64 KiB must be opted into
    {
        const void *bxs[10] = {reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC1, BQ_U8X>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_IH, BQ_U8X>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC1, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC1, BQ_U8>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_IH, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_IH, BQ_U8>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC2, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC2, BQ_U16>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC3, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC3, BQ_U16>)};
        for (const void *fn : bxs)
            UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, BX_LDS_BYTES));
        {
            const int wi_lds = (int)WI_LDS_BYTES; // 155,648: four transforms + the window
            UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wiener_istft_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, wi_lds));
            UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wiener_istft_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, wi_lds));
        }
#define UMX_GP_ATTR(MODE)                                                                                              \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 1, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(2, 2, 1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(2, 2, 2))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 1, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 2, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 2))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_pp_kernel<MODE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_pp_kernel<MODE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 2))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_ps_kernel<MODE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ps_lds_bytes(1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_ps_kernel<MODE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ps_lds_bytes(2)));
        UMX_GP_ATTR(G_FC1)
        UMX_GP_ATTR(G_IH)
        UMX_GP_ATTR(G_FC2)
        UMX_GP_ATTR(G_FC3)
#undef UMX_GP_ATTR
    }
    UMX_HIP_CHECK(hipDeviceSynchronize());
    return UMX_OK;
}

int umx_hip_ctx::ensure_staging()
{
    if (stage_in[0])
        return UMX_OK;
    UMX_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int si = 0; si < nslots; ++si)
    {
        for (hipEvent_t *e : {&slot[si].k_done, &slot[si].out_free})
            UMX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        if (int rc = dalloc(&stage_in[si], (size_t)2 * N * B, false))
            return rc;
        for (int k = 0; k < 4 * B; ++k)
            if (int rc = dalloc(&stage_out[si][k], (size_t)2 * N, false))
                return rc;
    }
    return UMX_OK;
}

// Host-pointer calls: where the stems of call k go out (round 3; kernel + copy timelines by tools/pcie_trace.sh).
//   * On the slot's own stream right behind its kernels (rounds 1-2), the download (48 ms for 32 lanes) stands in front of
//     call k + 2's upload and kernels; the two slots then fall into lock step -- kernels of two calls, downloads of two
//     calls, uploads of two calls, nothing overlapping: 128 ms per step against 78 ms of kernels.
//   * On the slot's own stream one call LATER (behind call k + 2's front stage): the same 129 ms.
//   * On the OTHER slot's stream behind call k + 1's kernels: 110-113 ms (the copies stay on the DMA engines, but a call's
//     kernels end at about the same time as the next call's, so the download still starts late).
//   * On a copy stream of its own behind an event (this code): 91-93 ms.  The runtime executes these copies as shader blits
//     (a download that does not follow kernels of its own stream), which wait for compute units behind the persistent LSTM
//     grids; a marker kernel in front of them does not change that.
// A separate UPLOAD stream as well made everything serial (141 ms): streams beyond the runtime's hardware queues share one.
int umx_hip_ctx::queue_download(const DeferredDownload &d, hipStream_t on)
{
    Slot &src = slot[d.si];
    if (on != src.stream)
        UMX_HIP_CHECK(hipStreamWaitEvent(on, src.k_done, 0));
    for (int ln = 0; ln < d.nb; ++ln)
        if (d.n[ln] > 0)
            for (int s2 = 0; s2 < 4; ++s2)
                UMX_HIP_CHECK(hipMemcpyAsync(d.host[4 * ln + s2], stage_out[d.si][4 * ln + s2], sizeof(float) * 2 * (size_t)d.n[ln],
                                             hipMemcpyDeviceToHost, on));
    UMX_HIP_CHECK(hipEventRecord(src.out_free, on));
    src.out_free_valid = true;
    return UMX_OK;
}

int umx_hip_ctx::sync_all()
{
    for (int si = 0; si < nslots; ++si)
        UMX_HIP_CHECK(hipStreamSynchronize(slot[si].stream));
    if (copy_stream)
        UMX_HIP_CHECK(hipStreamSynchronize(copy_stream));
    for (int si = 0; si < nslots; ++si)
        slot[si].out_free_valid = false; // drained
    return UMX_OK;
}
