// lstm_batch2.h -- the batched recurrence of lstm_batch.h for up to 48 track lanes: G = 2 or 3 groups of 16 lanes through
// one workgroup, in turn.
//
// A step of lstm_batch.h is a chain of latencies -- gate phase, the granules' way through the L2, the consumers' poll,
// the matrix instructions, a workgroup barrier that also absorbs the skew between the 32 workgroups of a chain -- of
// which the matrix pipe is busy for a fifth.  The matrix instruction is 16 tracks wide, so a second group of 16 lanes
// costs a second MFMA phase and a second gate phase, but they fall into the first group's waiting times:
//
//     waves 0..7  (poll + multiply, k-range w):   poll A | MFMA A | b0 | poll B | MFMA B | b1 | poll A' ...
//     waves 8..11 (gate phase, M tile w - 8):                       b0 | gates A, publish | b1 | gates B, publish | ...
//
// Group A's granules travel while the multiply waves work on group B and vice versa; nobody sleeps.  The weights in
// registers, the workgroup and its place in the chain are shared by both groups; per group: its own granule area, its own
// partial-sum buffer (single-buffered: it is written between the OTHER group's barrier and its own, read between its own
// and the other's), its own ring of W_ih x + b_ih rows.  Arithmetic per track is that of lstm_batch.h instruction for
// instruction (same fragments, same MFMA order, same fixed summation tree), so a track's bits do not depend on which
// kernel, group or lane it runs in (tests/test_gpu_batch.py).
// u8-resident W_hh only (one fp16 plane of q - 128 against two fp16 planes of h * 2^14); other weight forms run the
// 16-lane kernel once per group.
//
// Tried and measured in round 3, none of it faster than this form: requesting the NEXT turn's granules half a
// turn ahead, between the two halves of the current group's matrix instructions, so that the poll's L2 round trip would be
// off the turn (8.8 -> 10.7 ms per 32-lane launch, 12.8 -> 14.1 ms for 48 lanes: the granules of the slowest of the chain's
// 32 workgroups are not there yet, and a failed first attempt costs a second round trip; round 2 saw the same a whole turn
// ahead); sum_k h' from pair sums carried in the granules' free dword instead of the all-ones tile (round 3:
// matrix-pipe cycles -20 %, time +1.5 %: two dependent lane exchanges cost more than four queued matrix instructions).
// Kept: the W_ih-row ring at a pitch of 272 bytes (SQ_LDS_BANK_CONFLICT 1.5e8 -> 2e7 per launch; same time).
#pragma once
#include "lstm_batch.h"

namespace umx
{

constexpr int LSTMB2_THREADS = 768;
__host__ __device__ inline size_t lstmb2_lds_bytes(int groups, int bulk)
{
    return (size_t)groups * 8 * 16 * 16 * 16 /* part */ + (size_t)groups * 2 * bulk * 16 * LSTMB_RING_PITCH /* rings */ +
           (size_t)2 * groups * 8 * 16 * sizeof(float) /* sum_k h'_k per step parity, group, k-range and lane (fused row sums, lstm_batch.h) */;
}

template <int HL, int G, bool FAST, bool PRECISE>
__device__ __forceinline__ void lstmb2_body(const LstmBArgs &a, int chain, int slice, unsigned char *smem, int *abort_flag)
{
    constexpr int NKS = HL / 32, KSW = NKS >= 8 ? NKS / 8 : 1, NDW = NKS >= 8 ? 8 : NKS, NB = 16;
    const int target = a.tmap[chain >> 1], dir = chain & 1, wchain = target * 2 + dir;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 15, q = l >> 4;
    const int bulk = a.bulk, ring_mask = 2 * bulk - 1, T = a.T, S = a.S;
    const unsigned long long lane_mask = a.lane_mask;
    const bool dot_wave = w < NDW, gate_wave = w >= 8;
    const int gw = w - 8; // gate wave gw finishes M tile gw (units 4 gw .. 4 gw + 3 of the slice)

    float4 *part = reinterpret_cast<float4 *>(smem);                                // [G][8 waves][4 tiles][4 q][16]
    unsigned char *ring = smem + (size_t)G * 8 * 16 * NB * 16;                      // [G][2*bulk rows][16] blocks of 64 floats, LSTMB_RING_PITCH apart
    float *const hsw = reinterpret_cast<float *>(smem + lstmb2_lds_bytes(G, a.bulk) - (size_t)2 * G * 8 * 16 * sizeof(float)); // [2][G][8 waves][16]

    // ---- W_hh fragments (as lstm_batch.h, WQ form): lane (i = l & 15, q) of tile mt holds gate column 16 mt + i
    f16x8 Wf[4][KSW];
    if (dot_wave)
    {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
            {
                const size_t base = (((size_t)wchain * S + slice) * HL + (size_t)(w * KSW + ks) * 32 + 8 * q) * 64 + 16 * mt + n;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    Wf[mt][ks][j] = (_Float16)((float)a.Wq[base + (size_t)j * 64] - 128.0f);
            }
    }
    constexpr float HSCALE = 16384.0f;
    const f16x8 ones = __builtin_bit_cast(f16x8, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));
    const float wsc = a.wsc[wchain] * (1.0f / HSCALE), wof2 = (a.wof[wchain] + 128.0f * a.wsc[wchain]) * (1.0f / HSCALE);

    // ---- per group: lane activity, stream-state offsets, cell state of the gate lanes
    const int unit = slice * 16 + 4 * (w & 3) + q;
    bool lane_on[G];
    size_t st_h[G], st_c[G];
    float c[G], hlast[G];
    float4 bh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gate_wave)
        bh = *reinterpret_cast<const float4 *>(a.bhh + ((size_t)wchain * S + slice) * 64 + 4 * (4 * gw + q));
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
        lane_on[g] = (lane_mask >> (NB * g + n)) & 1ull;
        c[g] = 0.f;
        hlast[g] = 0.f;
        st_h[g] = (size_t)(NB * g + n) * a.state_stride + state_off(target, a.layer, dir, 0, HL);
        st_c[g] = (size_t)(NB * g + n) * a.state_stride + state_off(target, a.layer, dir, 1, HL);
        if (gate_wave && lane_on[g])
        {
            c[g] = a.state[st_c[g] + unit];
            hlast[g] = a.state[st_h[g] + unit];
        }
    }
    const size_t group_bytes = lstmb_granule_words(HL) * 4;
    const __amdgpu_buffer_rsrc_t gran_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.sync + LSTM_SYNC_HEADER_WORDS, 0, (int)(G * group_bytes), 0x00020000);
    gu32 *status = (gu32 *)a.status;
    const float *const Pp = a.P[target] + ((size_t)dir * S + slice) * 64 + l;
    float *const outp = a.out[target] + (size_t)n * a.out_stride + a.col0 + dir * HL + unit; // + NB g out_stride
    const size_t ldp = (size_t)a.ldp, ldo = (size_t)a.ldo, p_stride = a.p_stride;
    const unsigned tag_hi = a.tag_epoch << 12;
    const int t_begin = a.t_begin, t_end = a.t_end;
    // row sums of the layer's output from the all-ones tile (LstmBArgs::rs_dir, lstm_batch.h).  This kernel does NOT write the next
    // GEMM's A planes (LstmBArgs::planes is ignored: at twelve waves per workgroup there is no register left for it -- 39 spilled
    // with it); contexts of more than 32 lanes keep split_planes_kernel for the planes (the same bits) and take only the row sums here.
    float *const rsp = (a.rs_dir[target] && slice == 0) ? a.rs_dir[target] + (size_t)dir * a.rs_rows : nullptr;
    // row sum of the row that group g's turn of step sm multiplied with (lstm_batch.h, row_sum_of_step): by the last multiply wave,
    // in the shadow of its next poll for that group
    auto row_sum_of_step = [&](int g, int sm) {
        if (rsp && w == NDW - 1 && l < NB && sm > 0 && ((lane_mask >> (NB * g + l)) & 1ull))
        {
            float hp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < NDW; ++ww)
                hp[ww] = hsw[(((sm & 1) * G + g) * 8 + ww) * 16 + l];
            rsp[(size_t)(NB * g + l) * a.Tp + (size_t)(dir == 0 ? sm - 1 : T - sm)] = tree_sum<NDW>(hp) * (1.0f / 16384.0f);
        }
    };

    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    const unsigned ring_lds = (unsigned)(size_t)(lds_ptr)ring;
    auto fetch_rows = [&](int first_row) { // `bulk` rows of both groups, dealt to the multiply waves
        const int items = bulk * NB * G;
        for (int i = w; i < items; i += NDW)
        {
            const int g = i / (bulk * NB), r = first_row + (i % (bulk * NB)) / NB, nn = i % NB, ln = NB * g + nn;
            if (r < t_end && ((lane_mask >> ln) & 1ull))
                __builtin_amdgcn_global_load_lds((glb_ptr)(Pp + (size_t)ln * p_stride + (size_t)(dir == 0 ? r : T - 1 - r) * ldp),
                                                 (lds_ptr)(size_t)(ring_lds + (unsigned)LSTMB_RING_PITCH * (unsigned)(((g * 2 * bulk + (r & ring_mask)) * NB) + nn)), 4, 0, 0);
        }
    };
    if (dot_wave)
    {
        fetch_rows(t_begin);
        fetch_rows(t_begin + bulk);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
    }
    // vmcnt(0) for EVERY wave: what was loaded above (bias, state, weights) is then known to have arrived, and the compiler does
    // not place a "wait for everything" in front of the bias add of the gate phase -- which, inside the loop, is a wait for the
    // acknowledgements of the row / plane stores issued a few hundred cycles earlier, on the hand-off's critical path
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads(); // the first rows are read before the first barrier of the loop

    // in-kernel profiler (UMX_FLAG_LSTM_PROFILE; bench.py --lstm-profile): wave 0 (a multiply wave) and wave 8 (a gate wave) of
    // one workgroup count the shader cycles of a group's turn -- poll | fragments + matrix instructions + partial sums | barrier |
    // gate phase -- summed over groups and steps
    const bool prof = a.prof != nullptr && chain == 0 && slice == 0 && (w == 0 || w == 8);
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};

    for (int step = t_begin; step < t_end; ++step)
    {
        if (a.abort_at && step == a.abort_at && tid == 0)
        {
            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *abort_flag = 1;
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            unsigned prof_spins = 0;
            if (prof)
                c0 = c1 = clock64();
            if (dot_wave)
            {
                f16x8 hf[KSW][2];
                if (step == t_begin)
                {
                    // h_{t_begin - 1} from the fp32 stream state, split like a published granule
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks)
                    {
                        float hv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            hv[j] = lane_on[g] ? a.state[st_h[g] + (w * KSW + ks) * 32 + 8 * q + j] : 0.f;
                        uint4 p1, p2;
                        split2_f16(hv, HSCALE, p1, p2);
                        hf[ks][0] = __builtin_bit_cast(f16x8, p1);
                        hf[ks][1] = __builtin_bit_cast(f16x8, p2);
                    }
                }
                else
                {
                    // h_{step-1} of group g: granules of slot (step-1)&1 tagged `step`
                    const unsigned want = tag_hi | (unsigned)step;
                    int goff[KSW];
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks)
                        goff[ks] = (int)(g * group_bytes + lstmb_granule_index((step - 1) & 1, chain, (w * KSW + ks) * 32 + 8 * q, n, HL, NB) * 16);
                    uint4 v[KSW][4];
                    unsigned spins = 0;
                    for (;;)
                    {
                        bool ok = true;
                        if (lane_on[g])
                        {
#pragma unroll
                            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    v[ks][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(gran_rs, goff[ks] + i * 64 * NB, 0, 16)); // sc1
                            if (spins == 0)
                                row_sum_of_step(g, step - 1);
                            unsigned bad = 0;
#pragma unroll
                            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    bad |= v[ks][i].x ^ want;
                            ok = bad == 0;
                        }
                        if (__all(ok))
                            break;
                        if (++spins > LSTM_SPIN_LIMIT ||
                            ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                        {
                            if (l == 0)
                                __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            *abort_flag = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(LSTMB_RETRY_SLEEP);
                    }
                    prof_spins = spins;
                    if (prof)
                        c1 = clock64();
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks)
                    {
                        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                        const uint4 g0 = lane_on[g] ? v[ks][0] : z, g1 = lane_on[g] ? v[ks][1] : z, g2 = lane_on[g] ? v[ks][2] : z,
                                    g3 = lane_on[g] ? v[ks][3] : z;
                        hf[ks][0] = __builtin_bit_cast(f16x8, make_uint4(g0.y, g1.y, g2.y, g3.y));
                        hf[ks][1] = __builtin_bit_cast(f16x8, make_uint4(g0.z, g1.z, g2.z, g3.z));
                    }
                    // ring rows <= step - 2 may be replaced (every gate wave has read row step - 1 before the barriers of
                    // step - 1): rows [step-1+bulk, step-1+2 bulk) take the slots of [step-1-bulk, step-1); first read
                    // bulk - 1 steps from now, with this wave's next poll (vmcnt(0)) and a barrier in between
                    if (g == 0 && step - t_begin > bulk && ((step - t_begin) & (bulk - 1)) == (bulk > 1 ? 1 : 0))
                        fetch_rows(step - 1 + bulk);
                }
                floatx4 acc[4], accH = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ph = 1; ph >= 0; --ph) // smaller term first
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[mt][ks], hf[ks][ph], acc[mt], 0, 0, 0);
#pragma unroll
                for (int ph = 1; ph >= 0; --ph)
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks)
                        accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, hf[ks][ph], accH, 0, 0, 0);
                const float hs = wof2 * accH[0];
                if (rsp && q == 0) // sum over this wave's k-range of h'_{step-1}, group g, lane n
                    hsw[(((step & 1) * G + g) * 8 + w) * 16 + n] = accH[0];
                float4 *pw = part + ((size_t)((g * 8 + w) * 4) * 4 + q) * NB + n;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    pw[(size_t)mt * 4 * NB] = make_float4(wsc * acc[mt][0] + hs, wsc * acc[mt][1] + hs, wsc * acc[mt][2] + hs, wsc * acc[mt][3] + hs);
            }
            // the output row of the previous step goes out here, off the hand-off's critical path
            if (gate_wave && lane_on[g] && step > t_begin)
                outp[(size_t)(NB * g) * a.out_stride + (size_t)(dir == 0 ? step - 1 : T - step) * ldo] = hlast[g]; // lstm.cpp:163-164,170-171
            float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gate_wave)
                p4 = *reinterpret_cast<const float4 *>(ring + (size_t)((g * 2 * bulk + (step & ring_mask)) * NB + n) * LSTMB_RING_PITCH + 16 * (4 * gw + q));
            if (prof)
                c2 = clock64();
            __syncthreads();
            if (*abort_flag)
                return;
            if (prof)
                c3 = clock64();
            if (gate_wave)
            {
                float2v pa[8], pb[8];
#pragma unroll
                for (int ww = 0; ww < NDW; ++ww)
                {
                    const float4 v4 = part[((size_t)(((g * 8 + ww) * 4 + gw) * 4) + q) * NB + n];
                    pa[ww] = float2v{v4.x, v4.y};
                    pb[ww] = float2v{v4.z, v4.w};
                }
                const float2v sa = tree_sum2<NDW>(pa), sb = tree_sum2<NDW>(pb);
                const float s0 = sa.x, s1 = sa.y, s2 = sb.x, s3 = sb.y;
                // ((W_ih x + b_ih) + W_hh h) + b_hh, lstm.cpp:132-140
                const float pre_i = (p4.x + s0) + bh.x, pre_f = (p4.y + s1) + bh.y, pre_g = (p4.z + s2) + bh.z, pre_o = (p4.w + s3) + bh.w;
                float i_t, f_t, g_t, o_t;
                if (PRECISE)
                {
                    i_t = sigmoid_ref(pre_i);
                    f_t = sigmoid_ref(pre_f);
                    g_t = tanhf(pre_g);
                    o_t = sigmoid_ref(pre_o);
                }
                else
                {
                    i_t = __builtin_amdgcn_rcpf(1.0f + exp_hw(-pre_i));
                    f_t = __builtin_amdgcn_rcpf(1.0f + exp_hw(-pre_f));
                    g_t = tanh_hw(pre_g);
                    o_t = __builtin_amdgcn_rcpf(1.0f + exp_hw(-pre_o));
                }
                const float c_t = f_t * c[g] + i_t * g_t; // lstm.cpp:154-156
                const float h = o_t * (PRECISE ? tanhf(c_t) : tanh_hw(c_t)); // lstm.cpp:157
                const float hs14 = h * HSCALE;
                const _Float16 h1 = (_Float16)hs14, h2 = (_Float16)(hs14 - (float)h1);
                const unsigned b1 = __builtin_bit_cast(unsigned short, h1), b2 = __builtin_bit_cast(unsigned short, h2);
                const unsigned mine12 = b1 | (b2 << 16);
                const unsigned other12 = __builtin_amdgcn_permlane16_swap(mine12, mine12, false, false)[1]; // lane ^ 16 for the even rows (lstm_batch.h)
                if (lane_on[g])
                {
                    c[g] = c_t;
                    hlast[g] = h;
                    if ((q & 1) == 0) // publish the pair (this unit, the next), tagged step + 1
                    {
                        const uint4 gv = make_uint4(tag_hi | (unsigned)(step + 1), b1 | (other12 << 16), (mine12 >> 16) | (other12 & 0xffff0000u),
                                                    0u);
                        granule_store16<FAST>(gran_rs, (int)(g * group_bytes + lstmb_granule_index(step & 1, chain, unit, n, HL, NB) * 16), gv);
                    }
                }
            }
            if (prof)
            {
                const long long c4 = clock64();
                pc[0] += (unsigned long long)(c1 - c0);
                pc[1] += (unsigned long long)(c2 - c1);
                pc[2] += (unsigned long long)(c3 - c2);
                pc[3] += (unsigned long long)(c4 - c3);
                pc[4] += 1;
                pc[5] += prof_spins;
            }
        }
    }
    if (prof && l == 0)
    {
        const int pw_idx = w == 0 ? 0 : 1;
        for (int i = 0; i < 6; ++i)
            a.prof[(a.layer * 2 + pw_idx) * 8 + i] = (t_begin == 0 ? 0ull : a.prof[(a.layer * 2 + pw_idx) * 8 + i]) + pc[i];
        a.prof[(a.layer * 2 + pw_idx) * 8 + 6] = 0;
        a.prof[(a.layer * 2 + pw_idx) * 8 + 7] = 0;
    }
    if (t_end > t_begin)
#pragma unroll
        for (int g = 0; g < G; ++g)
            row_sum_of_step(g, t_end - 1);
    if (gate_wave) // lstm.cpp:160-161: the state carries into the next segment (and the next launch)
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (lane_on[g])
            {
                if (t_end > t_begin)
                    outp[(size_t)(NB * g) * a.out_stride + (size_t)(dir == 0 ? t_end - 1 : T - t_end) * ldo] = hlast[g];
                a.state_out[st_h[g] + unit] = hlast[g];
                a.state_out[st_c[g] + unit] = c[g];
            }
}

// grid = 8*S workgroups (1-D), 768 threads, plain launch; as lstm_batch_kernel
template <int HL, int G, bool PRECISE> __global__ __launch_bounds__(LSTMB2_THREADS) void lstm_batch2_kernel(LstmBArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lstmb2_smem[];
    __shared__ int s_ctl[4]; // chain, slice, fast, abort
    const int tid = threadIdx.x, S = a.S;
    if (tid == 0)
    {
        if (a.census)
            lstm_census(a.sync, a.status, S, (int)gridDim.x, a.force_safe, s_ctl);
        else
        {
            s_ctl[0] = (int)(blockIdx.x / S);
            s_ctl[1] = (int)(blockIdx.x % S);
            s_ctl[2] = 0;
            s_ctl[3] = 0;
        }
    }
    __syncthreads();
    const int chain = s_ctl[0], slice = s_ctl[1];
    if (s_ctl[3] || chain >= a.nchains)
        return;
    if (s_ctl[2])
        lstmb2_body<HL, G, true, PRECISE>(a, chain, slice, lstmb2_smem, &s_ctl[3]);
    else
        lstmb2_body<HL, G, false, PRECISE>(a, chain, slice, lstmb2_smem, &s_ctl[3]);
}

} // namespace umx
