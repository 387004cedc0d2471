// lstm_batch8.h -- the track-batched recurrence (lstm.cpp:101-179) sharded over LANES as well as over gate columns.
//
// lstm_batch.h shards a chain's 2048 gate columns over its workgroups and lets every workgroup serve ALL lanes of its group:
// each of them then reads the whole h of the chain for 16 lanes every step (65 KB of granules), the eight waves split the
// CONTRACTION (each polls its own k-range straight into matrix fragments) and meet through 80 KB of partial sums in LDS.  What a
// step costs there is exactly those two transfers (poll 1,350 + rendezvous 600-1,300 of ~5,600 cycles).
//
// Here a workgroup serves an OCTET of 8 lanes and owns 64 hidden units = 256 gate columns of its chain; a chain of a 32-lane
// launch is 8 column shards x 4 octets = 32 workgroups as before (one per CU, 256 in all), but
//   * a workgroup needs h of its own 8 lanes only: 512 units x 8 lanes = 2,048 granules = 32 KB per step, and a granule is read by
//     8 workgroups instead of 16: a quarter of lstm_batch2.h's hand-off bytes, half of lstm_batchs_kernel's;
//   * the matrix instruction's N = 16 is 8 lanes x the TWO fp16 planes of h * 2^14, so one v_mfma_f32_16x16x32_f16 multiplies both
//     planes; a wave owns 8 units = 32 gate columns = two M tiles for the WHOLE contraction (W_hh: 2 x 16 fragments = 128 VGPRs for
//     the layer), so its accumulators ARE the gate pre-activations: no partial sums, no second hand-over of 80 KB through LDS;
//   * h reaches the waves through LDS instead: the 512 threads poll four granules each, drop the payload dwords into fragment
//     order (16 KB per step), one LDS-only barrier, and every wave reads its B fragments with 16 conflict-free ds_read_b128.
// Per (unit, lane) and step:  W_hh h = (wsc 2^-14) [S1 + S2] + ((wof + 128 wsc) 2^-14) Hs, where S1 / S2 are the sums over k of
// (q_k - 128) h1_k / h2_k accumulated by the matrix pipe in k order (columns n and n + 8 of the result, added with one DPP
// rotation) and Hs = the sum of h' over the chain's units from an all-ones tile: wave w multiplies the k-steps of its eighth,
// the eight sums meet in LDS (32 bytes per lane) and are added in a fixed tree.  These are NOT the bits of lstm_batch.h (one
// accumulator per k-range there): a context uses one form or the other for every launch (engine_lstm.h), so a lane's result
// still never depends on which other lanes ride along; the same kernel one step per launch is its bit-identical per-step driver.
// Gate phase: every lane finishes ONE cell -- lanes n < 8 the unit of tile 0, lanes n >= 8 of tile 1 -- 64 cells per wave.
// Hand-off protocol, census, tags, bounded spins, fused A planes / row sums of the consuming GEMM: lstm_batch.h's.
#pragma once
#include "lstm_batch.h"

namespace umx
{

constexpr int LSTM8_TRACKS = 8;  // track lanes per workgroup
constexpr int LSTM8_UNITS = 64;  // hidden units per workgroup (8 per wave)
constexpr int LSTM8_OCTETS = 4;  // octets of a launch: 32 lanes
#define LSTM8_RETRY_SLEEP 1 // x64 cycles between failed polls

// bytes of one octet's granule area: [2 step slots][8 chains][Hl / 2 unit pairs][8 tracks] x 16 B
__host__ __device__ inline size_t lstm8_granule_bytes(int Hl) { return (size_t)2 * 8 * (Hl / 2) * LSTM8_TRACKS * 16; }
// LDS: h in fragment order [2 steps][Hl / 32 k-steps][4 k-groups][16 n] x 16 B, the eight k-range sums of h' [2][8 tracks][8 waves]
__host__ __device__ inline size_t lstm8_h_bytes(int Hl) { return (size_t)(Hl / 32) * 4 * 16 * 16; }
__host__ __device__ inline size_t lstm8_lds_bytes(int Hl) { return 2 * lstm8_h_bytes(Hl) + 2 * 8 * 8 * sizeof(float); }

template <int HL, bool FAST, bool PRECISE>
__device__ __forceinline__ void lstm8_body(const LstmBArgs &a, int chain, int shard, int octet, unsigned char *smem, int *abort_flag)
{
    constexpr int NKS = HL / 32;        // k-steps of the contraction
    constexpr int KSW = NKS / 8;        // k-steps whose sum of h' wave w forms
    constexpr int NLD = HL * 4 / 512;   // granules a thread polls per step
    constexpr int GPS = HL * 4;         // granules per (slot, chain): HL / 2 pairs x 8 tracks
    static_assert(HL % 256 == 0, "eight waves x 32-unit k-steps");
    const int target = a.tmap[chain >> 1], dir = chain & 1, wchain = target * 2 + dir;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 15, q = l >> 4;
    const int tr = n & 7, tile = n >> 3; // the cell this lane finishes: track tr of the octet, unit 4 tile + q of the wave's eight
    const int lane0 = a.lane_base + LSTM8_TRACKS * octet, T = a.T, S = a.S;
    const unsigned mask8 = (unsigned)(a.lane_mask >> lane0) & 0xffu;
    const bool lane_on = (mask8 >> tr) & 1u;
    const int U = shard * LSTM8_UNITS + w * 8 + tile * 4 + q; // hidden unit of the chain

    unsigned char *const hl = smem;                                                      // [2][NKS][4][16] x 16 B
    float *const hsp = reinterpret_cast<float *>(smem + 2 * lstm8_h_bytes(HL));          // [2][8 tracks][8 waves]

    // ---- W_hh fragments of the wave's two M tiles: lane (i = l & 15, q) holds gate column 16 mt + i (unit 4 mt + i / 4, gate i % 4), k = 32 ks + 8 q + j
    f16x8 Wf[2][NKS];
    {
        const unsigned char *wp[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
        {
            const int ug = shard * LSTM8_UNITS + w * 8 + mt * 4 + (n >> 2); // the weights stay in slices of 16 units ([chain][S][Hl][64])
            wp[mt] = a.Wq + (((size_t)wchain * S + (ug >> 4)) * HL + 8 * q) * 64 + 4 * (ug & 15) + (n & 3);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
        {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
            {
                f16x8 hw;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    hw[j] = (_Float16)((float)wp[mt][(size_t)(32 * ks + j) * 64] - 128.0f); // an integer in [-128, 127]: exact
                Wf[mt][ks] = hw;
            }
            asm volatile("" ::: "memory"); // (sixteen byte loads in flight, not 256: the prologue does not spill)
        }
    }
    constexpr float HSCALE = 16384.0f;
    const float wsc = a.wsc[wchain] * (1.0f / HSCALE), wof2 = (a.wof[wchain] + 128.0f * a.wsc[wchain]) * (1.0f / HSCALE);

    // ---- this lane's cell
    const size_t st_h = (size_t)(lane0 + tr) * a.state_stride + state_off(target, a.layer, dir, 0, HL);
    const size_t st_c = (size_t)(lane0 + tr) * a.state_stride + state_off(target, a.layer, dir, 1, HL);
    const float4 bh = *reinterpret_cast<const float4 *>(a.bhh + ((size_t)wchain * S + (U >> 4)) * 64 + 4 * (U & 15));
    float c = 0.f, hlast = 0.f;
    if (lane_on)
    {
        c = a.state[st_c + U];
        hlast = a.state[st_h + U];
    }
    unsigned plast = 0; // the fp16 planes of hlast (h1 | h2 << 16)

    // ---- what this thread polls: granule g = i 512 + tid of its (chain, octet): k-step g / 128, k-group (g / 32) % 4, pair (g / 8) % 4, track g % 8
    const int p_tr = tid & 7, p_pair = (tid >> 3) & 3, p_q = (tid >> 5) & 3, p_ks0 = tid >> 7;
    const bool p_on = (mask8 >> p_tr) & 1u;
    const int lds_w = ((p_ks0 * 4 + p_q) * 16 + p_tr) * 16 + p_pair * 4; // + i (4 x 1024) for load i; + 128 for the second plane
    const int t_begin = a.t_begin, t_end = a.t_end, poll_delay = a.poll_delay;
    // h_{t_begin - 1} from the fp32 stream state, split like a published granule; absent tracks are zero columns in both buffers
    {
        const size_t sh = (size_t)(lane0 + p_tr) * a.state_stride + state_off(target, a.layer, dir, 0, HL);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
        {
            const int k0 = (i * 4 + p_ks0) * 32 + p_q * 8 + p_pair * 2;
            unsigned d1 = 0u, d2 = 0u;
            if (p_on)
            {
                const float x0 = a.state[sh + k0] * HSCALE, x1 = a.state[sh + k0 + 1] * HSCALE;
                const _Float16 a1 = (_Float16)x0, b1 = (_Float16)x1;
                const _Float16 a2 = (_Float16)(x0 - (float)a1), b2 = (_Float16)(x1 - (float)b1);
                d1 = (unsigned)__builtin_bit_cast(unsigned short, a1) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                d2 = (unsigned)__builtin_bit_cast(unsigned short, a2) | ((unsigned)__builtin_bit_cast(unsigned short, b2) << 16);
            }
            unsigned char *dst = hl + (size_t)(t_begin & 1) * lstm8_h_bytes(HL) + lds_w + i * 4096;
            *reinterpret_cast<unsigned *>(dst) = d1;
            *reinterpret_cast<unsigned *>(dst + 128) = d2;
            if (!p_on)
            {
                unsigned char *other = hl + (size_t)((t_begin & 1) ^ 1) * lstm8_h_bytes(HL) + lds_w + i * 4096;
                *reinterpret_cast<unsigned *>(other) = 0u;
                *reinterpret_cast<unsigned *>(other + 128) = 0u;
            }
        }
    }

    const __amdgpu_buffer_rsrc_t gran_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char *>(a.sync + LSTM_SYNC_HEADER_WORDS) + (size_t)octet * lstm8_granule_bytes(HL), 0, (int)lstm8_granule_bytes(HL), 0x00020000);
    gu32 *status = (gu32 *)a.status;
    const size_t ldp = (size_t)a.ldp, ldo = (size_t)a.ldo;
    const float *const Pg = a.P[target] + (size_t)(lane0 + tr) * a.p_stride + ((size_t)dir * S + (U >> 4)) * 64 + 4 * (U & 15);
    float *const outp = a.out[target] + (size_t)(lane0 + tr) * a.out_stride + a.col0 + dir * HL + U;
    unsigned short *const plp = a.planes[target] ? a.planes[target] + (size_t)(lane0 + tr) * a.Tp * a.ldpl + a.col0 + dir * HL + U : nullptr;
    const size_t plane_elems = a.plane_elems, ldpl = (size_t)a.ldpl;
    // the row sum the consuming GEMM's affine fix-up needs = Hs of the step that multiplies with the row: one lane per track of the chain's first wave
    float *const rsp = (a.rs_dir[target] && shard == 0 && w == 0 && l < 8 && lane_on) ? a.rs_dir[target] + (size_t)dir * a.rs_rows + (size_t)(lane0 + tr) * a.Tp : nullptr;
    const unsigned tag_hi = a.tag_epoch << 12;
    const int gbase = chain * GPS * 16, gslot = 8 * GPS * 16; // bytes
    const int pub_off = gbase + (((U >> 3) * 4 + ((U & 7) >> 1)) * 8 + tr) * 16;
    float4 p4n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_on && t_begin < t_end)
        p4n = *reinterpret_cast<const float4 *>(Pg + (size_t)(dir == 0 ? t_begin : T - 1 - t_begin) * ldp);
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): weights, bias, state have arrived (no "wait for everything" inside the loop)
    const bool prof = a.prof != nullptr && octet == 0 && chain == 0 && shard == 0 && (w == 0 || w == LSTMB_PROF_WAVE);
    const int pw_idx = w == 0 ? 0 : 1;
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
    unsigned prof_spins = 0;
    const f16x8 ones16 = __builtin_bit_cast(f16x8, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));
#define LSTM8_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory") // LDS only: __syncthreads() is also a vmcnt(0) fence

    for (int step = t_begin; step < t_end; ++step)
    {
        long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (prof)
            c0 = clock64();
        unsigned char *const hb = hl + (size_t)(step & 1) * lstm8_h_bytes(HL);
        if (a.abort_at && step == a.abort_at && tid == 0)
        {
            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *abort_flag = 1;
        }
        if (step > t_begin)
        {
            // h_{step-1}: the granules of slot (step-1)&1 tagged `step`
            const unsigned want = tag_hi | (unsigned)step;
            const int goff = ((step - 1) & 1) * gslot + gbase + tid * 16;
            if (FAST)
                for (int d = poll_delay; d > 0; --d)
                    __builtin_amdgcn_s_sleep(1);
            uint4 v[NLD];
            unsigned spins = 0;
            for (;;)
            {
                bool ok = true;
                if (p_on)
                {
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                        v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(gran_rs, goff + i * 512 * 16, 0, 16)); // sc1
                    unsigned bad = 0;
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                        bad |= v[i].x ^ want;
                    ok = bad == 0;
                }
                if (__all(ok))
                    break;
                if (++spins > LSTM_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                {
                    if (l == 0)
                        __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *abort_flag = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(LSTM8_RETRY_SLEEP);
            }
            prof_spins = spins;
            if (p_on)
            {
#pragma unroll
                for (int i = 0; i < NLD; ++i)
                {
                    *reinterpret_cast<unsigned *>(hb + lds_w + i * 4096) = v[i].y;       // h1 of the pair
                    *reinterpret_cast<unsigned *>(hb + lds_w + i * 4096 + 128) = v[i].z; // h2 of the pair
                }
            }
        }
        // the output row of the PREVIOUS step goes out behind the polls (vector memory operations complete in order)
        if (lane_on && step > t_begin)
        {
            const size_t fr = (size_t)(dir == 0 ? step - 1 : T - step);
            if (!plp || a.write_f32)
                outp[fr * ldo] = hlast; // lstm.cpp:163-164,170-171
            if (plp)
            {
                plp[fr * ldpl] = (unsigned short)(plast & 0xffffu);
                plp[plane_elems + fr * ldpl] = (unsigned short)(plast >> 16);
            }
        }
        const float4 p4 = p4n; // row `step` of W_ih x + b_ih, requested a step ago
        if (lane_on && step + 1 < t_end)
            p4n = *reinterpret_cast<const float4 *>(Pg + (size_t)(dir == 0 ? step + 1 : T - 2 - step) * ldp);
        if (prof)
            c1 = clock64();
        LSTM8_LDS_BARRIER(); // h_{step-1} is in LDS
        if (*abort_flag)
            return;
        if (prof)
            c2 = clock64();

        // ---- matrix phase: the sum of h' over this wave's k-steps first (its LDS round trip hides behind the products)
        const unsigned char *const fb = hb + (q * 16 + n) * 16;
        floatx4 accH = {0.f, 0.f, 0.f, 0.f}, acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KSW; ++kk)
            accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones16, *reinterpret_cast<const f16x8 *>(fb + (w * KSW + kk) * 1024), accH, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
        {
            const f16x8 bf = *reinterpret_cast<const f16x8 *>(fb + ks * 1024);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][ks], bf, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][ks], bf, acc1, 0, 0, 0);
            if (ks == 3)
            {
                // (every row of accH holds the same sums; column n: plane n / 8 of track n % 8)
                const float hp = accH[0] + __int_as_float(dpp_row_ror<8>(__float_as_int(accH[0])));
                if (l < 8)
                    hsp[((step & 1) * 8 + l) * 8 + w] = hp;
            }
        }
        LSTM8_LDS_BARRIER(); // the eight k-range sums of h' are in LDS
        if (prof)
            c3 = clock64();

        // ---- gate phase: one cell per lane
        {
            const float4 ha = *reinterpret_cast<const float4 *>(hsp + ((step & 1) * 8 + tr) * 8), hc = *reinterpret_cast<const float4 *>(hsp + ((step & 1) * 8 + tr) * 8 + 4);
            const float hp8[8] = {ha.x, ha.y, ha.z, ha.w, hc.x, hc.y, hc.z, hc.w};
            const float Hs = tree_sum<8>(hp8);
            if (rsp && step > 0) // the row that this step multiplied with (step 0 multiplies with the carried state, not a row)
                rsp[(size_t)(dir == 0 ? step - 1 : T - step)] = Hs * (1.0f / 16384.0f);
            const float hterm = wof2 * Hs;
            float s[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                // lanes n < 8 finish tile 0: own column (plane 0) + column n + 8 (plane 1); lanes n >= 8 tile 1: own (plane 1) + column n - 8
                const float mine = tile ? acc1[r] : acc0[r], theirs = tile ? acc0[r] : acc1[r];
                const float sum = mine + __int_as_float(dpp_row_ror<8>(__float_as_int(theirs)));
                s[r] = wsc * sum + hterm;
            }
            // ((W_ih x + b_ih) + W_hh h) + b_hh, lstm.cpp:132-140
            const float pre_i = (p4.x + s[0]) + bh.x, pre_f = (p4.y + s[1]) + bh.y, pre_g = (p4.z + s[2]) + bh.z, pre_o = (p4.w + s[3]) + bh.w;
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
            const float c_t = f_t * c + i_t * g_t;                        // lstm.cpp:154-156
            const float h = o_t * (PRECISE ? tanhf(c_t) : tanh_hw(c_t)); // lstm.cpp:157
            const float hs14 = h * HSCALE;
            const _Float16 h1 = (_Float16)hs14, h2 = (_Float16)(hs14 - (float)h1);
            const unsigned b1 = __builtin_bit_cast(unsigned short, h1), b2 = __builtin_bit_cast(unsigned short, h2);
            const unsigned mine12 = b1 | (b2 << 16);
            // the odd unit of the pair sits 16 lanes up: row r + 1 into row r
            const unsigned other12 = __builtin_amdgcn_permlane16_swap(mine12, mine12, false, false)[1];
            if (lane_on)
            {
                c = c_t;
                hlast = h;
                plast = mine12;
                if ((q & 1) == 0) // publish the pair (this unit, the next), tagged step + 1
                    granule_store16<FAST>(gran_rs, (step & 1) * gslot + pub_off,
                                          make_uint4(tag_hi | (unsigned)(step + 1), b1 | (other12 << 16), (mine12 >> 16) | (other12 & 0xffff0000u), 0u));
            }
        }
        if (prof)
        {
            const long long c4 = clock64();
            pc[0] += (unsigned long long)(c1 - c0);
            pc[1] += (unsigned long long)(c3 - c2);
            pc[2] += (unsigned long long)(c2 - c1);
            pc[3] += (unsigned long long)(c4 - c3);
            pc[4] += 1;
            pc[5] += prof_spins;
        }
    }
#undef LSTM8_LDS_BARRIER
    if (lane_on) // lstm.cpp:160-161: the state carries into the next segment (and the next launch)
    {
        if (t_end > t_begin)
        {
            const size_t fr = (size_t)(dir == 0 ? t_end - 1 : T - t_end);
            outp[fr * ldo] = hlast;
            if (plp)
            {
                plp[fr * ldpl] = (unsigned short)(plast & 0xffffu);
                plp[plane_elems + fr * ldpl] = (unsigned short)(plast >> 16);
            }
        }
        a.state_out[st_h + U] = hlast;
        a.state_out[st_c + U] = c;
    }
    if (prof && l == 0)
    {
        for (int i = 0; i < 6; ++i)
            a.prof[(a.layer * 2 + pw_idx) * 8 + i] = (t_begin == 0 ? 0ull : a.prof[(a.layer * 2 + pw_idx) * 8 + i]) + pc[i];
        a.prof[(a.layer * 2 + pw_idx) * 8 + 6] = 0;
        a.prof[(a.layer * 2 + pw_idx) * 8 + 7] = 0;
    }
}

// grid: persistent (census = 1) 8 chains x 32 workgroups -- every XCD must receive 32 (one per CU): ticket / (HL / 64) picks one of the
// XCD's virtual chains (octet, chain), ticket % (HL / 64) the column shard, so that a hand-off domain lives on ONE XCD;
// one step per launch (census = 0): static roles, grid = octets x chains of the launch x shards.
template <int HL, bool PRECISE> __global__ __launch_bounds__(LSTM_THREADS, 2) void lstm_batch8_kernel(LstmBArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lstm8_smem[];
    __shared__ int s_ctl[4]; // chain, shard (or ticket), fast, abort
    constexpr int NSH = HL / LSTM8_UNITS, G = 32 / NSH; // column shards per chain; virtual chains per XCD (= octets for hidden 512)
    const int tid = threadIdx.x;
    if (tid == 0)
    {
        if (a.census)
            lstm_census(a.sync, a.status, 32, (int)gridDim.x, a.force_safe, s_ctl);
        else
        {
            s_ctl[2] = 0;
            s_ctl[3] = 0;
        }
    }
    __syncthreads();
    if (s_ctl[3])
        return;
    int vc, shard; // virtual chain = octet x 8 + chain
    if (s_ctl[2])
    {
        vc = s_ctl[0] * G + s_ctl[1] / NSH;
        shard = s_ctl[1] % NSH;
    }
    else
    {
        const int nch = (int)gridDim.x / (G * NSH), v = (int)blockIdx.x / NSH;
        vc = (v / nch) * 8 + v % nch;
        shard = (int)blockIdx.x % NSH;
    }
    const int octet = vc >> 3, chain = vc & 7;
    const unsigned mask8 = (unsigned)(a.lane_mask >> (a.lane_base + LSTM8_TRACKS * octet)) & 0xffu;
    if (chain >= a.nchains || mask8 == 0u)
        return;
    if (s_ctl[2])
        lstm8_body<HL, true, PRECISE>(a, chain, shard, octet, lstm8_smem, &s_ctl[3]);
    else
        lstm8_body<HL, false, PRECISE>(a, chain, shard, octet, lstm8_smem, &s_ctl[3]);
}

} // namespace umx
