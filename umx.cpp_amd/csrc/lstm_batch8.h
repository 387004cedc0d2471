// lstm_batch8.h -- the track-batched recurrence (lstm.cpp:101-179) sharded over LANES as well as over gate columns.
//
// lstm_batch.h shards a chain's 2048 gate columns over its workgroups and lets every workgroup serve ALL lanes of its group:
// each of them then reads the whole h of the chain for 16 lanes every step (65 KB of granules), the eight waves split the
// CONTRACTION (each polls its own k-range straight into matrix fragments) and meet through 80 KB of partial sums in LDS.  What a
// step costs there is exactly those two transfers (poll 1,350 + rendezvous 600-1,300 of ~5,600 cycles).
//
// Here a workgroup serves an OCTET of 8 lanes and owns 64 hidden units = 256 gate columns of its chain; a chain of a 32-lane
// launch is 8 column shards x 4 octets = 32 workgroups as before (one per CU, 256 in all), but
//   * a workgroup needs h of its own 8 lanes only: 512 units x 8 lanes = 2,048 granules per step (12 of a granule's 16 bytes are
//     loaded: 24 KB), and a granule is read by 8 workgroups instead of 16: under half of the hand-off bytes of round 4's side-by-side kernel;
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
// What a step costs NOW is the CU's one vector-memory address path: the instruction count per step is what the
// structure below minimises -- 12-byte polls, the output row's planes staged in LDS and stored by two waves, row sums by one.
// Hand-off protocol, census, tags, bounded spins, fused A planes / row sums of the consuming GEMM: lstm_batch.h's.
#pragma once
#include "lstm_batch.h"

namespace umx
{

constexpr int LSTM8_TRACKS = 8;  // track lanes per workgroup
constexpr int LSTM8_UNITS = 64;  // hidden units per workgroup (8 per wave)
// octets side by side in a launch = the virtual chains an XCD holds beside the 8 real ones: 32 workgroups per XCD / column shards per chain
__host__ __device__ constexpr int lstm8_octets(int Hl) { return 32 / (Hl / LSTM8_UNITS); } // hidden 1024: 4 (32 lanes), hidden 512: 8 (64 lanes)
#define LSTM8_RETRY_SLEEP 1 // x64 cycles between failed polls
#define LSTM8_OOR 0x7ffffff0 // a buffer offset beyond every resource of this kernel: the store is dropped

// bytes of one octet's granule area: [2 step slots][8 chains][Hl / 2 unit pairs][8 tracks] x 16 B
__host__ __device__ inline size_t lstm8_granule_bytes(int Hl) { return (size_t)2 * 8 * (Hl / 2) * LSTM8_TRACKS * 16; }
// LDS: h in fragment order [2 steps][Hl / 32 k-steps][4 k-groups][16 n] x 16 B, the eight k-range sums of h' [2][8 tracks][8 waves]
__host__ __device__ inline size_t lstm8_h_bytes(int Hl) { return (size_t)(Hl / 32) * 4 * 16 * 16; }
// the row's planes staged for their store: [2 planes][8 tracks] rows of 64 units at a pitch of 72 ushorts = 36 banks: the eight tracks of a
// gate wave land on eight different bank quads (at 64 they shared one: 8-way conflicts, 224 of 240 conflict cycles per step and workgroup
// in profiles/r05_v4_pmc_sq_counters.txt), and the 16-byte reads of the storing waves stay aligned (144 = 9 x 16)
#ifndef LSTM8_STG_PITCH
#define LSTM8_STG_PITCH 72
#endif
__host__ __device__ inline size_t lstm8_lds_bytes(int Hl, int no = 1) { return (size_t)no * (2 * lstm8_h_bytes(Hl) + 2 * 8 * 8 * sizeof(float) + 2 * 8 * LSTM8_STG_PITCH * 2); } // no: octets per workgroup

// NO = octets a workgroup serves IN TURN (2: launches of 33 .. 64 lanes -- octet o and octet o + 4 with the same weight fragments).
// A step of one octet is a dependent chain  publication -> L2 -> polls (a round of loads ~1,100 cycles + ~700 until the last wave's
// have come through the CU's one path) -> matrix phase -> gate phase, of which only the last two keep the workgroup's pipes busy
// (profiles/r05_lstm_batch8_first.txt).  With two octets the polls of the NEXT turn are issued inside the matrix phase of the current
// one -- its granules were published a whole turn ago -- and checked when that turn begins: the hand-off of one octet runs under the
// matrix and gate phases of the other.  Per (unit, lane) the arithmetic does not know about turns: the bits of NO = 1.
#ifndef LSTM8_EARLY_KS
#define LSTM8_EARLY_KS 13 // the next turn's polls are issued behind this k-step of the matrix phase (-1: in front of it)
#endif
#ifndef LSTM8_EXPERIMENT
#define LSTM8_EXPERIMENT 0 // timing builds only (wrong results): 1 = no request for the next row of W_ih x, 2 = no output stores inside the loop
#endif
#ifndef LSTM8_STAGE_PLANES
#define LSTM8_STAGE_PLANES 1 // the row's fused A planes go through LDS and leave as two 16-byte store instructions per turn (0: two 2-byte stores per lane)
#endif
#ifndef LSTM8_PIN_ORDER
#define LSTM8_PIN_ORDER 1 // the matrix phase in the order written (sched_barrier): two matrix instructions, then the fragment read LSTM8_FRAG_AHEAD k-steps ahead -- left to itself the scheduler sinks every read to its use (1,655 -> 1,540 cycles)
#endif
#ifndef LSTM8_FRAG_AHEAD
#define LSTM8_FRAG_AHEAD 4 // h fragments a wave reads ahead of its matrix instructions (2 / 4 / 8: the compiler's schedule, and the time, are the same;
                           // forcing the read-ahead into the schedule with sched_group_barrier: matrix phase 1,655 -> 1,790-1,860 cycles)
#endif
template <int HL, bool FAST, bool PRECISE, int NO>
__device__ __forceinline__ void lstm8_body(const LstmBArgs &a, int chain, int shard, int octet0, unsigned char *smem, int *abort_flag)
{
    constexpr int NKS = HL / 32;        // k-steps of the contraction
    constexpr int KSW = NKS / 8;        // k-steps whose sum of h' wave w forms
    constexpr int NLD = HL * 4 / 512;   // granules a thread polls per step
    constexpr int GPS = HL * 4;         // granules per (slot, chain): HL / 2 pairs x 8 tracks
    constexpr int HB = (HL / 32) * 4 * 16 * 16;         // = lstm8_h_bytes(HL)
    constexpr int OCT = lstm8_octets(HL); // octets of a launch (NO = 2: the workgroup of octet o also serves octet o + OCT)
    constexpr int EARLY = LSTM8_EARLY_KS < 0 ? 0 : (LSTM8_EARLY_KS < NKS ? LSTM8_EARLY_KS : NKS - 3); // k-step behind which the next turn's polls go out
    constexpr int HS_KS = 3 < NKS ? 3 : NKS - 1; // k-step behind which a wave's k-range sum of h' is written (its KSW products are long done)
    static_assert(HL % 256 == 0 && (NO == 1 || NO == 2) && 32 % (HL / LSTM8_UNITS) == 0 && NKS >= 8, "eight waves x 32-unit k-steps; one octet or two in turn");
    const int target = a.tmap[chain >> 1], dir = chain & 1, wchain = target * 2 + dir;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 15, q = l >> 4;
    const int tr = n & 7, tile = n >> 3; // the cell this lane finishes: track tr of the octet, unit 4 tile + q of the wave's eight
    const int T = a.T, S = a.S;
    const int U = shard * LSTM8_UNITS + w * 8 + tile * 4 + q; // hidden unit of the chain

    unsigned char *const hl = smem;                                                        // [NO][2][NKS][4][16] x 16 B
    float *const hsp = reinterpret_cast<float *>(smem + (size_t)NO * 2 * HB);              // [NO][2][8 tracks][8 waves]
    unsigned short *const stg = reinterpret_cast<unsigned short *>(smem + (size_t)NO * (2 * HB + 2 * 8 * 8 * sizeof(float))); // [NO][2 planes][8 tracks][LSTM8_STG_PITCH]

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
            asm volatile("" ::: "memory"); // (sixteen byte loads in flight, not 256)
        }
    }
    constexpr float HSCALE = 16384.0f;
    const float wsc = a.wsc[wchain] * (1.0f / HSCALE), wof2 = (a.wof[wchain] + 128.0f * a.wsc[wchain]) * (1.0f / HSCALE);
    const float4 bh = *reinterpret_cast<const float4 *>(a.bhh + ((size_t)wchain * S + (U >> 4)) * 64 + 4 * (U & 15));

    // ---- what this thread polls: granule g = i 512 + tid of its (chain, octet): k-step g / 128, k-group (g / 32) % 4, pair (g / 8) % 4, track g % 8
    const int p_tr = tid & 7, p_pair = (tid >> 3) & 3, p_q = (tid >> 5) & 3, p_ks0 = tid >> 7;
    const int lds_w = ((p_ks0 * 4 + p_q) * 16 + p_tr) * 16 + p_pair * 4; // + i (4 x 1024) for load i; + 128 for the second plane
    const int t_begin = a.t_begin, t_end = a.t_end, poll_delay = a.poll_delay;
    const size_t ldp = (size_t)a.ldp, ldo = (size_t)a.ldo, plane_elems = a.plane_elems, ldpl = (size_t)a.ldpl;
    const unsigned tag_hi = a.tag_epoch << 12;
    const int gbase = chain * GPS * 16, gslot = 8 * GPS * 16; // bytes
    const int pub_off = gbase + (((U >> 3) * 4 + ((U & 7) >> 1)) * 8 + tr) * 16;
    gu32 *status = (gu32 *)a.status;

    // ---- per octet: this lane's cell, its rows, the granule area
    bool on[NO], lane_on[NO], p_on[NO];
    unsigned mask8s[NO];
    float c[NO], hlast[NO];
    unsigned plast[NO]; // the fp16 planes of hlast (h1 | h2 << 16)
    float4 p4n[NO];
    const float *Pg[NO];
    float *outp[NO];
    int rs_off[NO]; // byte offset of this lane's row sums (LSTM8_OOR: none)
    unsigned short *plp[NO];
    __amdgpu_buffer_rsrc_t gran_rs[NO];
    int lane0[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o)
    {
        const int octet = octet0 + o * OCT;
        lane0[o] = a.lane_base + LSTM8_TRACKS * octet;
        const unsigned mask8 = (unsigned)(a.lane_mask >> lane0[o]) & 0xffu;
        on[o] = mask8 != 0u;
        mask8s[o] = mask8;
        lane_on[o] = (mask8 >> tr) & 1u;
        p_on[o] = (mask8 >> p_tr) & 1u;
        const size_t st = (size_t)(lane0[o] + tr) * a.state_stride;
        c[o] = lane_on[o] ? a.state[st + state_off(target, a.layer, dir, 1, HL) + U] : 0.f;
        hlast[o] = lane_on[o] ? a.state[st + state_off(target, a.layer, dir, 0, HL) + U] : 0.f;
        plast[o] = 0u;
        Pg[o] = a.P[target] + (size_t)(lane0[o] + tr) * a.p_stride + ((size_t)dir * S + (U >> 4)) * 64 + 4 * (U & 15);
        outp[o] = a.out[target] + (size_t)(lane0[o] + tr) * a.out_stride + a.col0 + dir * HL + U;
        plp[o] = a.planes[target] ? a.planes[target] + (size_t)(lane0[o] + tr) * a.Tp * a.ldpl + a.col0 + dir * HL + U : nullptr;
        // the row sum the consuming GEMM's affine fix-up needs = Hs of the step that multiplies with the row: one lane per track of the chain's first wave
        rs_off[o] = (shard == 0 && w == 0 && l < 8 && lane_on[o]) ? (int)(((size_t)dir * a.rs_rows + (size_t)(lane0[o] + tr) * a.Tp) * 4) : LSTM8_OOR;
        gran_rs[o] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(a.sync + LSTM_SYNC_HEADER_WORDS) + (size_t)octet * lstm8_granule_bytes(HL), 0,
                                                       (int)lstm8_granule_bytes(HL), 0x00020000);
        p4n[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on[o] && t_begin < t_end)
            p4n[o] = stream_load4(Pg[o] + (size_t)(dir == 0 ? t_begin : T - 1 - t_begin) * ldp);
        // h_{t_begin - 1} from the fp32 stream state, split like a published granule; absent tracks are zero columns in both buffers
        const size_t sh = (size_t)(lane0[o] + p_tr) * a.state_stride + state_off(target, a.layer, dir, 0, HL);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
        {
            const int k0 = (i * 4 + p_ks0) * 32 + p_q * 8 + p_pair * 2;
            unsigned d1 = 0u, d2 = 0u;
            if (p_on[o])
            {
                const float x0 = a.state[sh + k0] * HSCALE, x1 = a.state[sh + k0 + 1] * HSCALE;
                const _Float16 a1 = (_Float16)x0, b1 = (_Float16)x1;
                const _Float16 a2 = (_Float16)(x0 - (float)a1), b2 = (_Float16)(x1 - (float)b1);
                d1 = (unsigned)__builtin_bit_cast(unsigned short, a1) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                d2 = (unsigned)__builtin_bit_cast(unsigned short, a2) | ((unsigned)__builtin_bit_cast(unsigned short, b2) << 16);
            }
            unsigned char *dst = hl + (size_t)(o * 2 + (t_begin & 1)) * HB + lds_w + i * 4096;
            *reinterpret_cast<unsigned *>(dst) = d1;
            *reinterpret_cast<unsigned *>(dst + 128) = d2;
            if (!p_on[o])
            {
                unsigned char *other = hl + (size_t)(o * 2 + ((t_begin & 1) ^ 1)) * HB + lds_w + i * 4096;
                *reinterpret_cast<unsigned *>(other) = 0u;
                *reinterpret_cast<unsigned *>(other + 128) = 0u;
            }
        }
    }
    // Publication and row sums go out as buffer stores that EVERY lane issues (lanes with nothing to store: an offset beyond the
    // resource, dropped by the range check): no branch around them, so the compiler knows how many memory operations follow the polls
    // issued during a turn, and the wait in front of their check does not include these stores' acknowledgements.
    const __amdgpu_buffer_rsrc_t rs_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.rs_dir[target], 0, a.rs_dir[target] ? (int)((size_t)2 * a.rs_rows * 4) : 0, 0x00020000);
    const bool rs_wave = shard == 0 && w == 0, have_planes = a.planes[target] != nullptr;
    const bool both_on = NO == 2 && on[0] && on[NO - 1]; // the turns overlap each other's hand-off only if there are two
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): weights, bias, state have arrived (no "wait for everything" inside the loop)
    const bool prof = a.prof != nullptr && octet0 == 0 && chain == 0 && shard == 0 && (w == 0 || w == LSTMB_PROF_WAVE);
    const int pw_idx = w == 0 ? 0 : 1;
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, pc6 = 0, pc7 = 0;
    unsigned prof_spins = 0;
    const f16x8 ones16 = __builtin_bit_cast(f16x8, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));
#define LSTM8_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory") // LDS only: __syncthreads() is also a vmcnt(0) fence
    typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
    v3u32 v[NLD];         // the polls of one turn: {tag, h1 pair, h2 pair} = the first 12 bytes of a granule (its fourth dword is unused: a
                          // register nobody reads would be handed out again while the load is still in flight -- and waited for)
    bool pending = false; // ... issued during the turn before
    const int goff0 = gbase + tid * 16;
#define LSTM8_POLL(o_, step_)                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) v[i] =                                                                                \
        __builtin_amdgcn_raw_buffer_load_b96(gran_rs[o_], (((step_)-1) & 1) * gslot + goff0 + i * 512 * 16, 0, 16) /* sc1 */

    for (int step = t_begin; step < t_end; ++step)
    {
        if (a.abort_at && step == a.abort_at && tid == 0)
        {
            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *abort_flag = 1;
        }
#pragma unroll
        for (int o = 0; o < NO; ++o)
        {
            if (NO > 1 && !on[o])
                continue;
            long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, ca = 0;
            if (prof)
                ca = c0 = clock64();
            unsigned char *const hb = hl + (size_t)(o * 2 + (step & 1)) * HB;
            float *const hs = hsp + (o * 2 + (step & 1)) * 64;
            if (step > t_begin)
            {
                // h_{step-1}: the granules of slot (step-1)&1 tagged `step`
                const unsigned want = tag_hi | (unsigned)step;
                if (FAST && !pending)
                    for (int d = poll_delay; d > 0; --d)
                        __builtin_amdgcn_s_sleep(1);
                unsigned spins = 0;
                auto tags_bad = [&]() {
                    unsigned bad = 0;
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                        bad |= v[i].x ^ want;
                    return bad;
                };
                // two code paths on purpose: the wait in front of a check covers everything issued before it on ANY path into it, and
                // behind polls issued a turn ago sit that turn's publication and row-sum stores -- whose acknowledgements are not needed here
                bool ok = true;
                if (pending)
                {
                    if (p_on[o])
                        ok = tags_bad() == 0u;
                }
                else if (p_on[o])
                {
                    LSTM8_POLL(o, step);
                    ok = tags_bad() == 0u;
                }
                pending = false;
                while (!__all(ok))
                {
                    if (++spins > LSTM_SPIN_LIMIT || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                    {
                        if (l == 0)
                            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *abort_flag = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(LSTM8_RETRY_SLEEP);
                    if (p_on[o])
                    {
                        LSTM8_POLL(o, step);
                        ok = tags_bad() == 0u;
                    }
                }
                prof_spins = spins;
                if (prof)
                    ca = clock64(); // (the check is through: the polls are there)
                if (p_on[o])
                {
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                    {
                        *reinterpret_cast<unsigned *>(hb + lds_w + i * 4096) = v[i].y;       // h1 of the pair
                        *reinterpret_cast<unsigned *>(hb + lds_w + i * 4096 + 128) = v[i].z; // h2 of the pair
                    }
                }
            }
            // the output row of the PREVIOUS step and the request for the next row of W_ih x go out behind the polls (vector memory
            // operations complete in order).  (Behind the barrier instead, under the matrix phase: the ~300 cycles their issue takes on
            // the CU's one address path then stall the wave's matrix instructions for longer -- 5.85 against 5.5 ms per 32-lane launch.)
            // (The row's two planes as ONE dword store per lane -- the even unit of a pair writing (h1, h1') into plane 0, the odd unit
            // (h2, h2') into plane 1 -- saves a store instruction and costs more in the gate phase: 9.7 against 9.4 ms per 64-lane launch.)
            if (!(LSTM8_EXPERIMENT & 2) && lane_on[o] && step > t_begin)
            {
                const size_t fr = (size_t)(dir == 0 ? step - 1 : T - step);
                if (!plp[o] || a.write_f32)
                    outp[o][fr * ldo] = hlast[o]; // lstm.cpp:163-164,170-171
                if (plp[o] && !LSTM8_STAGE_PLANES)
                {
                    plp[o][fr * ldpl] = (unsigned short)(plast[o] & 0xffffu);
                    plp[o][plane_elems + fr * ldpl] = (unsigned short)(plast[o] >> 16);
                }
            }
            const float4 p4 = p4n[o]; // row `step` of W_ih x + b_ih, requested a step ago
            if (!(LSTM8_EXPERIMENT & 1) && lane_on[o] && step + 1 < t_end)
                p4n[o] = stream_load4(Pg[o] + (size_t)(dir == 0 ? step + 1 : T - 2 - step) * ldp); // (a row of W_ih x + b_ih is read once)
            if (prof)
                c1 = clock64();
            LSTM8_LDS_BARRIER(); // h_{step-1} is in LDS
            if (*abort_flag)
                return;
            if (prof)
                c2 = clock64();
            // the PREVIOUS step's row of the fused A planes: staged in LDS by the gate lanes (two bytes each), it leaves as ONE 16-byte store
            // per lane of two waves -- a whole 128-byte line per (track, plane) -- instead of two 2-byte stores per lane of all eight:
            // 2 instead of 16 instructions per turn on the CU's one address path
            if (LSTM8_STAGE_PLANES && !(LSTM8_EXPERIMENT & 2) && have_planes && step > t_begin && w < 2)
            {
                const int strk = l >> 3, sgrp = l & 7;
                if ((mask8s[o] >> strk) & 1u)
                {
                    const uint4 val = *reinterpret_cast<const uint4 *>(stg + ((o * 2 + w) * 8 + strk) * LSTM8_STG_PITCH + sgrp * 8);
                    unsigned short *dst = a.planes[target] + (size_t)w * plane_elems + ((size_t)(lane0[o] + strk) * a.Tp + (size_t)(dir == 0 ? step - 1 : T - step)) * ldpl +
                                          a.col0 + dir * HL + shard * LSTM8_UNITS + sgrp * 8;
                    stream_store4u(reinterpret_cast<uint4 *>(dst), val);
                }
            }
            // the next turn's polls: the other octet's granules were published a turn ago
            const int no = NO - 1 - o, nstep = o + 1 < NO ? step : step + 1;
            const bool issue_next = NO > 1 && both_on && nstep > t_begin && nstep < t_end;

            // ---- matrix phase: ONE basic block (the k-range sum of h' is stored by every lane -- all 64 hold the sum of their column's
            // track -- and the next turn's polls are loads every lane issues, out of range where there is nothing to poll), fragments
            // read LSTM8_FRAG_AHEAD k-steps ahead of the products that take them
            const unsigned char *const fb = hb + (q * 16 + n) * 16;
            floatx4 accH = {0.f, 0.f, 0.f, 0.f}, acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            f16x8 bf[LSTM8_FRAG_AHEAD];
#pragma unroll
            for (int i = 0; i < LSTM8_FRAG_AHEAD; ++i)
                bf[i] = *reinterpret_cast<const f16x8 *>(fb + i * 1024);
            if (LSTM8_PIN_ORDER)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KSW; ++kk)
                accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones16, *reinterpret_cast<const f16x8 *>(fb + (w * KSW + kk) * 1024), accH, 0, 0, 0);
            const int poll_base = (NO > 1 && issue_next && p_on[no]) ? ((nstep - 1) & 1) * gslot + goff0 : LSTM8_OOR;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
            {
                const f16x8 cur = bf[ks % LSTM8_FRAG_AHEAD];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][ks], cur, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][ks], cur, acc1, 0, 0, 0);
                if (ks + LSTM8_FRAG_AHEAD < NKS)
                    bf[ks % LSTM8_FRAG_AHEAD] = *reinterpret_cast<const f16x8 *>(fb + (ks + LSTM8_FRAG_AHEAD) * 1024);
                if (LSTM8_PIN_ORDER)
                    __builtin_amdgcn_sched_barrier(0); // the order as written: two matrix instructions, the read LSTM8_FRAG_AHEAD k-steps ahead
                if (ks == HS_KS) // (every row of accH holds the same sums; column n: plane n / 8 of track n % 8)
                    hs[tr * 8 + w] = accH[0] + __int_as_float(dpp_row_ror<8>(__float_as_int(accH[0])));
                if (NO > 1 && ks == EARLY)
                {
#pragma unroll
                    for (int i = 0; i < NLD; ++i)
                        v[i] = __builtin_amdgcn_raw_buffer_load_b96(gran_rs[no], poll_base == LSTM8_OOR ? LSTM8_OOR : poll_base + i * 512 * 16, 0, 16); // sc1
                    pending = issue_next;
                }
            }
            LSTM8_LDS_BARRIER(); // the eight k-range sums of h' are in LDS
            if (prof)
                c3 = clock64();

            // ---- gate phase: one cell per lane
            {
                const float4 ha = *reinterpret_cast<const float4 *>(hs + tr * 8), hc = *reinterpret_cast<const float4 *>(hs + tr * 8 + 4);
                const float hp8[8] = {ha.x, ha.y, ha.z, ha.w, hc.x, hc.y, hc.z, hc.w};
                const float Hs = tree_sum<8>(hp8);
                // the row that this step multiplied with (step 0 multiplies with the carried state, not a row)
                if (rs_wave) // (wave-uniform: one wave of a chain)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(Hs * (1.0f / 16384.0f)), rs_rs,
                                                          (rs_off[o] != LSTM8_OOR && step > 0) ? rs_off[o] + (dir == 0 ? step - 1 : T - step) * 4 : LSTM8_OOR, 0, 0);
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
                const float c_t = f_t * c[o] + i_t * g_t;                     // lstm.cpp:154-156
                const float h = o_t * (PRECISE ? tanhf(c_t) : tanh_hw(c_t)); // lstm.cpp:157
                const float hs14 = h * HSCALE;
                const _Float16 h1 = (_Float16)hs14, h2 = (_Float16)(hs14 - (float)h1);
                const unsigned b1 = __builtin_bit_cast(unsigned short, h1), b2 = __builtin_bit_cast(unsigned short, h2);
                const unsigned mine12 = b1 | (b2 << 16);
                // the odd unit of the pair sits 16 lanes up: row r + 1 into row r
                const unsigned other12 = __builtin_amdgcn_permlane16_swap(mine12, mine12, false, false)[1];
                if (lane_on[o])
                {
                    c[o] = c_t;
                    hlast[o] = h;
                    plast[o] = mine12;
                }
                if (LSTM8_STAGE_PLANES && have_planes)
                {
                    // two bytes per lane and plane, rows of LSTM8_STG_PITCH ushorts: the eight tracks of a wave on eight bank quads.  (One
                    // dword per pair of units from the even unit's lane -- half the lanes under an exec mask -- measured 2.5 % slower
                    // per launch: profiles/r06_lstm8_staging_ab.txt.)
                    unsigned short *sg = stg + (o * 2 * 8 + tr) * LSTM8_STG_PITCH + w * 8 + tile * 4 + q;
                    sg[0] = (unsigned short)b1;
                    sg[8 * LSTM8_STG_PITCH] = (unsigned short)b2;
                }
                // the even unit of a pair publishes it (this unit, the next), tagged step + 1
                granule_store16<FAST>(gran_rs[o], (lane_on[o] && (q & 1) == 0) ? (step & 1) * gslot + pub_off : LSTM8_OOR,
                                      make_uint4(tag_hi | (unsigned)(step + 1), b1 | (other12 << 16), (mine12 >> 16) | (other12 & 0xffff0000u), 0u));
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
                pc6 += (unsigned long long)(ca - c0);
                pc7 += (unsigned long long)(c1 - ca);
            }
        }
    }
#undef LSTM8_LDS_BARRIER
#undef LSTM8_POLL
#pragma unroll
    for (int o = 0; o < NO; ++o)
        if (lane_on[o]) // lstm.cpp:160-161: the state carries into the next segment (and the next launch)
        {
            if (t_end > t_begin)
            {
                const size_t fr = (size_t)(dir == 0 ? t_end - 1 : T - t_end);
                outp[o][fr * ldo] = hlast[o];
                if (plp[o])
                {
                    plp[o][fr * ldpl] = (unsigned short)(plast[o] & 0xffffu);
                    plp[o][plane_elems + fr * ldpl] = (unsigned short)(plast[o] >> 16);
                }
            }
            const size_t st = (size_t)(lane0[o] + tr) * a.state_stride;
            a.state_out[st + state_off(target, a.layer, dir, 0, HL) + U] = hlast[o];
            a.state_out[st + state_off(target, a.layer, dir, 1, HL) + U] = c[o];
        }
    if (prof && l == 0)
    {
        for (int i = 0; i < 6; ++i)
            a.prof[(a.layer * 2 + pw_idx) * 8 + i] = (t_begin == 0 ? 0ull : a.prof[(a.layer * 2 + pw_idx) * 8 + i]) + pc[i];
        a.prof[(a.layer * 2 + pw_idx) * 8 + 6] = pc6; // until the polls' check is through
        a.prof[(a.layer * 2 + pw_idx) * 8 + 7] = pc7; // LDS writes, the previous row's stores, the next row's request
    }
}

// grid: persistent (census = 1) 8 chains x 32 workgroups -- every XCD must receive 32 (one per CU): ticket / (HL / 64) picks one of the
// XCD's virtual chains (octet, chain), ticket % (HL / 64) the column shard, so that a hand-off domain lives on ONE XCD;
// one step per launch (census = 0): static roles, grid = octets x chains of the launch x shards.  NO = 2: the workgroup of octet o also
// serves octet o + lstm8_octets (hidden 1024: launches of 33 .. 64 lanes; hidden 512 has its 64 lanes side by side).
template <int HL, bool PRECISE, int NO> __global__ __launch_bounds__(LSTM_THREADS, 2) void lstm_batch8_kernel(LstmBArgs a)
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
    unsigned mask = (unsigned)(a.lane_mask >> (a.lane_base + LSTM8_TRACKS * octet)) & 0xffu;
    if (NO > 1)
        mask |= (unsigned)(a.lane_mask >> (a.lane_base + LSTM8_TRACKS * (octet + lstm8_octets(HL)))) & 0xffu;
    if (chain >= a.nchains || mask == 0u)
        return;
    if (s_ctl[2])
        lstm8_body<HL, true, PRECISE, NO>(a, chain, shard, octet, lstm8_smem, &s_ctl[3]);
    else
        lstm8_body<HL, false, PRECISE, NO>(a, chain, shard, octet, lstm8_smem, &s_ctl[3]);
}

} // namespace umx
