// split.cpp -- segmented apply: split_inference / shift_inference (host side of umx_host.h).
// Follows umx.cpp:99-295 over a pluggable per-segment backend (the HIP engine in the product,
// the oracle in CPU tests).  One deliberate deviation, declared in the header: sum_weight is
// zero-initialised over its whole length.
#include "../../include/umx_host.h"
#include "shard_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace
{
void seterr(char *err, const std::string &m)
{
    if (err)
        snprintf(err, UMX_ERRLEN, "%s", m.c_str());
}
const float kOverlap = 0.25f; // inference.hpp:15
} // namespace

extern "C" float umx_transition_weight(int k, int chunk_len, int segment_samples)
{
    // umx.cpp:197-206: weight[i] = weight[N-1-i] = i+1 for i < N/2, / max, ^1.0; used as
    // weight(k % chunk_length) (umx.cpp:246)
    const int N = segment_samples;
    const int i = k % chunk_len;
    const float raw = (float)((i < N / 2) ? i + 1 : N - i);
    const float mx = (float)(N / 2);
    return std::pow(raw / mx, 1.0f);
}

extern "C" int umx_segment_plan(int length, int segment_samples, int *offsets, int *lengths, int cap)
{
    const int stride = (int)((1 - kOverlap) * segment_samples); // umx.cpp:181
    int n = 0;
    if (stride <= 0)
        return 0;
    for (long long offset = 0; offset < length; offset += stride) // umx.cpp:214
    {
        if (n < cap)
        {
            if (offsets)
                offsets[n] = (int)offset;
            if (lengths)
                lengths[n] = std::min(segment_samples, length - (int)offset); // umx.cpp:217
        }
        ++n;
    }
    return n;
}

extern "C" int umx_split_inference(const umx_backend *be, const float *audio, int length, int segment_samples,
                                   float *const out[4], void (*progress)(float, void *), void *progress_user,
                                   char *err)
{
    if (!be || !be->segment || !audio || !out || length < 1 || segment_samples < 2)
    {
        seterr(err, "umx_split_inference: bad argument");
        return UMX_ERR_ARG;
    }
    const int N = segment_samples;
    const int stride = (int)((1 - kOverlap) * N);
    if (be->reset) // umx.cpp:167-171: the 4 lstm_data are created (zeroed) once per track
        if (int rc = be->reset(be->user))
        {
            seterr(err, "backend reset failed");
            return rc;
        }
    std::vector<float> sum_w((size_t)length, 0.0f);
    for (int t = 0; t < 4; ++t)
        std::fill(out[t], out[t] + (size_t)2 * length, 0.0f); // umx.cpp:186-195
    const float total_reps = std::ceil((float)length / (float)stride); // umx.cpp:208
    float done = 0.f;
    std::vector<float> chunk[4];
    for (long long off = 0; off < length; off += stride)
    {
        const int offset = (int)off;
        const int chunk_len = std::min(N, length - offset);
        float *co[4];
        for (int t = 0; t < 4; ++t)
        {
            chunk[t].assign((size_t)2 * chunk_len, 0.0f);
            co[t] = chunk[t].data();
        }
        if (int rc = be->segment(be->user, audio + (size_t)2 * offset, chunk_len, co)) // umx.cpp:226-227
        {
            seterr(err, "segment backend failed at offset " + std::to_string(offset));
            return rc ? rc : UMX_HOST_ERR_BACKEND;
        }
        done += 1.0f / total_reps; // umx.cpp:229
        if (progress)
            progress(done, progress_user);
        for (int k = 0; k < N && offset + k < length; ++k) // umx.cpp:234-260
        {
            const float w = umx_transition_weight(k, chunk_len, N);
            for (int t = 0; t < 4; ++t)
            {
                out[t][2 * (size_t)(offset + k)] += w * chunk[t][2 * (size_t)k];
                out[t][2 * (size_t)(offset + k) + 1] += w * chunk[t][2 * (size_t)k + 1];
            }
            sum_w[offset + k] += w;
        }
    }
    for (int t = 0; t < 4; ++t) // umx.cpp:264-273
        for (int k = 0; k < length; ++k)
        {
            out[t][2 * (size_t)k] /= sum_w[k];
            out[t][2 * (size_t)k + 1] /= sum_w[k];
        }
    return UMX_OK;
}

extern "C" int umx_shift_inference(const umx_backend *be, const float *audio, int length, int segment_samples,
                                   int offset, float *const out[4], void (*progress)(float, void *),
                                   void *progress_user, char *err)
{
    if (!be || !audio || !out || length < 1)
    {
        seterr(err, "umx_shift_inference: bad argument");
        return UMX_ERR_ARG;
    }
    const int max_shift = UMX_MAX_SHIFT_SAMPLES; // umx.cpp:112-113
    if (offset < 0)
        offset = UMX_REFERENCE_SHIFT; // umx.cpp:115: rand() % 22050, never seeded in the reference (see umx_hip.h)
    if (offset >= max_shift)
    {
        seterr(err, "shift offset must be < 22050");
        return UMX_ERR_ARG;
    }
    // umx.cpp:120-122 sizes the buffer length + max_shift - offset and then writes [offset, offset + length): past
    // its end for offset > max_shift / 2 (undefined behaviour in the reference; its unseeded rand() always gives 4033).
    // Same size wherever the reference is defined, large enough everywhere else.
    const int L2 = length + std::max(max_shift - offset, offset);
    std::vector<float> shifted((size_t)2 * L2, 0.0f);
    memcpy(shifted.data() + (size_t)2 * offset, audio, sizeof(float) * 2 * (size_t)length);
    std::vector<float> full[4];
    float *fo[4];
    for (int t = 0; t < 4; ++t)
    {
        full[t].resize((size_t)2 * L2);
        fo[t] = full[t].data();
    }
    if (int rc = umx_split_inference(be, shifted.data(), L2, segment_samples, fo, progress, progress_user, err))
        return rc;
    for (int t = 0; t < 4; ++t) // umx.cpp:136-147
        memcpy(out[t], full[t].data() + (size_t)2 * offset, sizeof(float) * 2 * (size_t)length);
    return UMX_OK;
}

// ---------------------------------------------------------------- one track over several ranks (exact carry mode)
extern "C" int umx_split_inference_carry(const umx_phased_backend *be, const umx_p2p *p2p, int rank, int world,
                                         const float *audio, int length, int segment_samples, float *const out[4], char *err)
{
    if (!be || !be->begin || !be->layer || !be->end || !be->get_layer || !be->set_layer || !audio || length < 1 ||
        segment_samples < 2 || world < 1 || rank < 0 || rank >= world || (world > 1 && (!p2p || !p2p->send || !p2p->recv)) ||
        (rank == 0 && !out))
    {
        seterr(err, "umx_split_inference_carry: bad argument");
        return UMX_ERR_ARG;
    }
    const int N = segment_samples, stride = (int)((1 - kOverlap) * N); // umx.cpp:181
    std::vector<int> offsets;
    for (long long off = 0; off < length; off += stride) // umx.cpp:214
        offsets.push_back((int)off);
    const int nseg = (int)offsets.size();
    const size_t nf = be->layer_floats;
    std::vector<float> st(nf);
    std::vector<std::vector<float>> mine(nseg); // this rank's weighted stems, [4][2][n] per segment
    for (int i = rank; i < nseg; i += world)
    {
        const int off = offsets[i], n = std::min(N, length - off); // umx.cpp:217
        if (int rc = be->begin(be->user, audio + (size_t)2 * off, n))
        {
            seterr(err, "phased backend: begin failed");
            return rc;
        }
        for (int l = 0; l < 3; ++l)
        {
            int rc = UMX_OK;
            if (i == 0) // umx.cpp:167-171 / lstm.cpp:82: the track starts from zero state
            {
                std::fill(st.begin(), st.end(), 0.0f);
                rc = be->set_layer(be->user, l, st.data());
            }
            else if (world > 1)
            {
                rc = p2p->recv(p2p->user, st.data(), nf, (i - 1) % world);
                if (!rc)
                    rc = be->set_layer(be->user, l, st.data());
            } // world == 1: the state segment i-1 left is already in place
            if (!rc)
                rc = be->layer(be->user, l);
            if (!rc && world > 1 && i + 1 < nseg)
            {
                rc = be->get_layer(be->user, l, st.data());
                if (!rc)
                    rc = p2p->send(p2p->user, st.data(), nf, (i + 1) % world);
            }
            if (rc)
            {
                seterr(err, "carry driver: layer " + std::to_string(l) + " of segment " + std::to_string(i) + " failed");
                return rc;
            }
        }
        std::vector<float> stems[4];
        float *so[4];
        for (int t = 0; t < 4; ++t)
        {
            stems[t].assign((size_t)2 * n, 0.0f);
            so[t] = stems[t].data();
        }
        if (int rc = be->end(be->user, so))
        {
            seterr(err, "phased backend: end failed");
            return rc;
        }
        mine[i].resize((size_t)4 * 2 * n);
        for (int t = 0; t < 4; ++t)
            for (int k = 0; k < n; ++k)
            {
                const float w = umx_transition_weight(k, n, N); // umx.cpp:246
                mine[i][((size_t)t * n + k) * 2] = w * stems[t][2 * (size_t)k];
                mine[i][((size_t)t * n + k) * 2 + 1] = w * stems[t][2 * (size_t)k + 1];
            }
    }
    // gather on rank 0 in segment order (= the reference's accumulation order), umx.cpp:234-273
    if (rank != 0)
    {
        for (int i = rank; i < nseg; i += world)
            if (int rc = p2p->send(p2p->user, mine[i].data(), mine[i].size(), 0))
            {
                seterr(err, "carry driver: sending stems failed");
                return rc;
            }
        return UMX_OK;
    }
    std::vector<float> sum_w((size_t)length, 0.0f), buf;
    for (int t = 0; t < 4; ++t)
        std::fill(out[t], out[t] + (size_t)2 * length, 0.0f);
    for (int i = 0; i < nseg; ++i)
    {
        const int off = offsets[i], n = std::min(N, length - off);
        const float *ws = mine[i].data();
        if (i % world != 0)
        {
            buf.resize((size_t)4 * 2 * n);
            if (int rc = p2p->recv(p2p->user, buf.data(), buf.size(), i % world))
            {
                seterr(err, "carry driver: receiving stems failed");
                return rc;
            }
            ws = buf.data();
        }
        for (int k = 0; k < n; ++k)
        {
            for (int t = 0; t < 4; ++t)
            {
                out[t][2 * (size_t)(off + k)] += ws[((size_t)t * n + k) * 2];
                out[t][2 * (size_t)(off + k) + 1] += ws[((size_t)t * n + k) * 2 + 1];
            }
            sum_w[off + k] += umx_transition_weight(k, n, N);
        }
    }
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < length; ++k)
        {
            out[t][2 * (size_t)k] /= sum_w[k];
            out[t][2 * (size_t)k + 1] /= sum_w[k];
        }
    return UMX_OK;
}

// ---------------------------------------------------------------- one track over several ranks, by source model x segment
extern "C" int umx_split_inference_targets(const umx_target_backend *be, const umx_p2p *p2p, int rank, int world, const float *audio,
                                           int length, int segment_samples, float *const out[4], char *err)
{
    if (!be || !be->begin || !be->layer || !be->get_state || !be->set_state || !be->masks || !be->get_mag || !be->set_mag ||
        !be->finish || !be->discard || !audio || length < 1 || segment_samples < 2 || world < 1 || rank < 0 || rank >= world ||
        (world > 1 && (!p2p || !p2p->send || !p2p->recv)) || (rank == 0 && !out))
    {
        seterr(err, "umx_split_inference_targets: bad argument");
        return UMX_ERR_ARG;
    }
    const umx_plan::Plan pl = umx_plan::make_plan(world, true);
    const int N = segment_samples, stride = (int)((1 - kOverlap) * N); // umx.cpp:181
    std::vector<int> offsets;
    for (long long off = 0; off < length; off += stride) // umx.cpp:214
        offsets.push_back((int)off);
    const int nseg = (int)offsets.size();
    const size_t nf = be->target_layer_floats, nm = be->mag_floats;
    unsigned mask = 0;
    std::vector<int> mine_t;
    for (int t = 0; t < 4; ++t)
        if (pl.owns_target(rank, t))
        {
            mask |= 1u << t;
            mine_t.push_back(t);
        }
    std::vector<float> st(nf), mag(nm);
    std::vector<std::vector<float>> mine(nseg); // weighted stems of the segments this rank filters, [4][n][2]
#define UMX_TRY(expr, what)                                                                                          \
    if (int rc_ = (expr))                                                                                            \
    {                                                                                                                \
        seterr(err, std::string("target driver: ") + what + " failed (segment " + std::to_string(i) + ")");         \
        return rc_;                                                                                                  \
    }
    for (int i = 0; i < nseg; ++i)
    {
        if (!pl.runs_segment(rank, i))
            continue;
        const int off = offsets[i], n = std::min(N, length - off); // umx.cpp:217
        UMX_TRY(be->begin(be->user, audio + (size_t)2 * off, n, mask), "begin");
        for (int l = 0; l < 3; ++l)
        {
            for (int t : mine_t)
            {
                if (i == 0) // umx.cpp:167-171 / lstm.cpp:82: the track starts from zero state
                {
                    std::fill(st.begin(), st.end(), 0.0f);
                    UMX_TRY(be->set_state(be->user, l, t, st.data()), "set_state");
                }
                else if (pl.P > 1)
                {
                    UMX_TRY(p2p->recv(p2p->user, st.data(), nf, pl.prev_rank(rank, i)), "receiving LSTM state");
                    UMX_TRY(be->set_state(be->user, l, t, st.data()), "set_state");
                } // P == 1: the state segment i-1 left is already in place
            }
            UMX_TRY(be->layer(be->user, l), "LSTM layer");
            if (pl.P > 1 && i + 1 < nseg)
                for (int t : mine_t)
                {
                    UMX_TRY(be->get_state(be->user, l, t, st.data()), "get_state");
                    UMX_TRY(p2p->send(p2p->user, st.data(), nf, pl.next_rank(rank, i)), "sending LSTM state");
                }
        }
        UMX_TRY(be->masks(be->user), "masks");
        const int wr = pl.wiener_rank(i);
        if (wr != rank)
        {
            for (int t : mine_t)
            {
                UMX_TRY(be->get_mag(be->user, t, mag.data()), "get_mag");
                UMX_TRY(p2p->send(p2p->user, mag.data(), nm, wr), "sending target magnitudes");
            }
            UMX_TRY(be->discard(be->user), "discard");
            continue;
        }
        for (int t = 0; t < 4; ++t)
            if (!pl.owns_target(rank, t))
            {
                UMX_TRY(p2p->recv(p2p->user, mag.data(), nm, pl.owner_of_target(t, pl.stage(rank))), "receiving target magnitudes");
                UMX_TRY(be->set_mag(be->user, t, mag.data()), "set_mag");
            }
        std::vector<float> stems[4];
        float *so[4];
        for (int t = 0; t < 4; ++t)
        {
            stems[t].assign((size_t)2 * n, 0.0f);
            so[t] = stems[t].data();
        }
        UMX_TRY(be->finish(be->user, so), "finish"); // wiener_filter + istft, inference.cpp:192-207
        mine[i].resize((size_t)4 * 2 * n);
        for (int t = 0; t < 4; ++t)
            for (int k = 0; k < n; ++k)
            {
                const float w = umx_transition_weight(k, n, N); // umx.cpp:246
                mine[i][((size_t)t * n + k) * 2] = w * stems[t][2 * (size_t)k];
                mine[i][((size_t)t * n + k) * 2 + 1] = w * stems[t][2 * (size_t)k + 1];
            }
    }
    // gather on rank 0 in segment order (= the reference's accumulation order), umx.cpp:234-273
    if (rank != 0)
    {
        for (int i = 0; i < nseg; ++i)
            if (pl.wiener_rank(i) == rank)
                UMX_TRY(p2p->send(p2p->user, mine[i].data(), mine[i].size(), 0), "sending stems");
        return UMX_OK;
    }
    std::vector<float> sum_w((size_t)length, 0.0f), buf;
    for (int t = 0; t < 4; ++t)
        std::fill(out[t], out[t] + (size_t)2 * length, 0.0f);
    for (int i = 0; i < nseg; ++i)
    {
        const int off = offsets[i], n = std::min(N, length - off);
        const float *ws = mine[i].data();
        if (pl.wiener_rank(i) != 0)
        {
            buf.resize((size_t)4 * 2 * n);
            UMX_TRY(p2p->recv(p2p->user, buf.data(), buf.size(), pl.wiener_rank(i)), "receiving stems");
            ws = buf.data();
        }
        for (int k = 0; k < n; ++k)
        {
            for (int t = 0; t < 4; ++t)
            {
                out[t][2 * (size_t)(off + k)] += ws[((size_t)t * n + k) * 2];
                out[t][2 * (size_t)(off + k) + 1] += ws[((size_t)t * n + k) * 2 + 1];
            }
            sum_w[off + k] += umx_transition_weight(k, n, N);
        }
    }
#undef UMX_TRY
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < length; ++k)
        {
            out[t][2 * (size_t)k] /= sum_w[k];
            out[t][2 * (size_t)k + 1] /= sum_w[k];
        }
    return UMX_OK;
}
