// lstm_wavefront.h -- one persistent kernel that advances up to three LSTM layers of three CONSECUTIVE
// segments at once:  R_0(segment k), R_1(segment k-1), R_2(segment k-2)   (Hl = 512 only).
//
// Why: one layer of one segment is 2584 serially dependent steps, and a step's floor is the cross-CU
// hand-off of h (~0.25 us) plus the serial gate phase -- the CUs are idle most of the time.  The exact
// wavefront of SURVEY 8e (layer l of segment s+1 only needs layer l of segment s to be finished) provides
// three INDEPENDENT chain sets in steady state.  Interleaving them inside each workgroup hides the
// hand-off of one set behind the arithmetic of the other two: in steady state the kernel is bound by VALU
// issue, not by latency.  Arithmetic per chain is exactly that of lstm_persistent_kernel / lstm_step_kernel
// (same DPP-rotation order, same partial-sum tree, same gates), so results are bitwise identical.
//
// STATUS (round 1): opt-in (UMX_PIPELINE=wavefront).  Measured on MI355X it is bitwise exact in lock-step
// form (WF_TASK_BARRIER=1) but NOT faster than the two-slot pipeline of engine.hip (13.0 vs 10.9 ms per
// 60 s segment): with 250 VGPRs only two waves fit a SIMD and every task still pays one L2 round trip for
// its poll, so the latency is not hidden as planned.  Kept as the basis for a later redesign.
//
// Structure of a workgroup (chain c, slice j), 8 waves:
//   every wave w : for each active set m: poll the 64 granules of its k-range (the load was issued one
//                  task earlier), 64 v_fmac over the set's register-resident W_hh slice (3 x 64 VGPRs),
//                  4 partial sums -> LDS, LDS arrival counter += 1
//   wave m       : additionally owns the gates of set m: waits for the 8 arrivals of (m, step) on the LDS
//                  counter, adds the partials, applies the gates, publishes the 16 new granules.
// Waves run decoupled (a wave that waits for a granule does not stop the others), which is what lets the
// two waves sharing a SIMD fill each other's stalls.
#pragma once
#include "lstm_kernels.h"

#ifndef WF_NO_FUSE
#define WF_NO_FUSE 0 // debugging: gates right after the set's own MACs instead of fused into the next task
#endif
#ifndef WF_TASK_BARRIER
#define WF_TASK_BARRIER 1 // __syncthreads() after every task (lock-step waves).  REQUIRED for exact results:
// with fully decoupled waves (0) the kernel is ~8 % faster but intermittently starts a chain from a wrong
// value (tools/wfrace.py; error largest at step 0 and decaying) -- an unresolved hazard, so 0 is for
// experiments only
#endif
#ifndef WF_NO_PREFETCH
#define WF_NO_PREFETCH 0 // debugging: no poll issued one task ahead
#endif

namespace umx
{

// LDS-typed pointers: a generic pointer to the abort flag turns into flat_load, after which hipcc can no
// longer count vmcnt and drains every prefetched poll with vmcnt(0)
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) unsigned lds_u32;

struct LstmSet
{
    const float *W;   // [chains][S][Hl][64] of this set's layer
    const float *bhh; // [chains][S][64]
    const float *P[4];
    float *out[4];
    int ldo, col0, layer, active;
    unsigned gran_off; // first granule (u64 index) of this set inside the sync buffer's granule area
};

struct LstmWaveArgs
{
    LstmSet set[3];
    float *state;
    unsigned *sync;   // [0..7] census, [8] arrivals, granules from LSTM_SYNC_HEADER_WORDS
    unsigned *status; // [0] abort/timeout, [1] fast flag
    int S, T, ldp, nchains;
    int tmap[4];
    int force_safe;
    unsigned tag_base; // granule tag of step s = tag_base + s + 1 (unique per launch, see LstmArgs)
    unsigned long long *prof; // optional: [wave 0 | wave 3][poll, dot, gate_wait, gate+dot, tasks]
};

constexpr int LSTM_WF_HL = 512, LSTM_WF_KPW = 64;

// MASK = bitmask of the active sets (compile time, so every register array index is static and the
// hand-over order is known): 1, 3 while the wavefront fills; 7 in steady state; 6, 4 while it drains.
constexpr int wf_next(int mask, int m)
{
    int q = m;
    for (int i = 0; i < 3; ++i)
    {
        q = (q + 1) % 3;
        if ((mask >> q) & 1)
            return q;
    }
    return m;
}
constexpr int wf_wrap(int mask, int m) { return wf_next(mask, m) <= m ? 1 : 0; }

constexpr int wf_prev(int mask, int m)
{
    int q = m;
    for (int i = 0; i < 3; ++i)
    {
        q = (q + 2) % 3;
        if ((mask >> q) & 1)
            return q;
    }
    return m;
}

// everything one wave keeps across steps
struct WfWave
{
    WSlice Wd[3];
    float hval[3];
    unsigned long long pend[3]; // granule loads issued one task ahead
    float c, bh, hlast, p, p_old; // gate duty: cell state, b_hh, last h, P of this step / of the previous step
    unsigned long long pc[5];
    bool prof;
};

// the 64 MACs of one wave for (set M, step) and its 4 partial sums -> LDS + arrival
template <int M>
__device__ __forceinline__ void wf_dot(WfWave &ws, int step, int w, int l, float (*part)[2][8][64], lds_u32 *cnt)
{
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    DotDpp<15>::run(ws.Wd[M], __float_as_int(ws.hval[M]), acc);
    float s0, s1;
    {
        const auto r01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0]), __float_as_uint(acc[1]), false, false);
        const auto r23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[2]), __float_as_uint(acc[3]), false, false);
        s0 = __uint_as_float(r01[0]) + __uint_as_float(r01[1]);
        s1 = __uint_as_float(r23[0]) + __uint_as_float(r23[1]);
    }
    const float t0 = s0 + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s0), 0x401F));
    const float t1 = s1 + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(s1), 0x401F));
    if ((l & 16) == 0)
    {
        float *pp = &part[M][step & 1][w][4 * (l & 15) + (l >> 5)];
        pp[0] = t0;
        pp[2] = t1;
    }
    // partials before the arrival: LDS executes one wave's operations in order; the wait also keeps the
    // compiler from moving the atomic up (no vmcnt wait here: the prefetched poll stays in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (l == 0)
        __hip_atomic_fetch_add(&cnt[M], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// wait until all 8 waves have delivered their partials of (set G, gstep)
template <int G>
__device__ __forceinline__ bool wf_wait_partials(const LstmWaveArgs &a, int gstep, int l, lds_u32 *cnt, lds_i32 *abort_flag)
{
    const unsigned need = 8u * (unsigned)(gstep + 1);
    unsigned spins = 0;
    while (__hip_atomic_load(&cnt[G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
    {
        if (++spins > LSTM_SPIN_LIMIT || ((spins & 255u) == 0 && *(volatile lds_i32 *)abort_flag))
        {
            if (l == 0)
                __hip_atomic_store((gu32 *)a.status, 0x40000000u + (unsigned)gstep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(volatile lds_i32 *)abort_flag = 1;
            return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

// gates of (set G, gstep) by its gate wave: add the 8 partials, apply the gates, publish 16 granules
template <int G, bool FAST, bool PRECISE>
__device__ __forceinline__ void wf_gates(const LstmWaveArgs &a, WfWave &ws, int gstep, float p, int chain, int target, int dir,
                                         int unit, int l, float (*part)[2][8][64])
{
    constexpr int Hl = LSTM_WF_HL;
    gu64 *gran = (gu64 *)(a.sync + LSTM_SYNC_HEADER_WORDS);
    const int tg = dir == 0 ? gstep : a.T - 1 - gstep;
    float(*pq)[64] = part[G][gstep & 1];
    const float s = ((pq[0][l] + pq[1][l]) + (pq[2][l] + pq[3][l])) + ((pq[4][l] + pq[5][l]) + (pq[6][l] + pq[7][l]));
    const float pre = (p + s) + ws.bh;
    float h;
    lstm_cell<PRECISE>(pre, l, ws.c, h);
    if ((l & 3) == 0)
    {
        const unsigned long long gv =
            ((unsigned long long)(a.tag_base + (unsigned)(gstep + 1)) << 32) | (unsigned long long)__float_as_uint(h);
        granule_store<FAST>(gran + a.set[G].gran_off + granule_index(gstep & 1, chain, unit, a.S), gv);
        a.set[G].out[target][(size_t)tg * a.set[G].ldo + a.set[G].col0 + dir * Hl + unit] = h;
        ws.hlast = h;
    }
}

// One task = (set M, step): poll h, prefetch the next task's poll, 64 MACs.  The gates of a set are a
// ~650-cycle dependent chain on ONE wave; executed on their own they would delay that wave's partials for
// the following sets, and those delays chain around the three sets (3 x (gates + dot) per step).  So the
// gate wave G = wf_prev(M) runs the gates of (G, .) FUSED with its own MACs of set M, in one basic block:
// the independent v_fmac stream fills the latency gaps of the gate chain.  With a single active set the
// next task depends on these very gates, so they run right after the MACs instead.
// Returns false when the launch is being aborted.
template <int MASK, int M, bool FAST, bool PRECISE>
__device__ __forceinline__ bool wf_task(const LstmWaveArgs &a, WfWave &ws, int step, int chain, int target, int dir,
                                        int unit, bool gate_duty, float (*part)[2][8][64], lds_u32 *cnt, lds_i32 *abort_flag)
{
    constexpr int KPW = LSTM_WF_KPW;
    if (!((MASK >> M) & 1))
        return true;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    gu64 *gran = (gu64 *)(a.sync + LSTM_SYNC_HEADER_WORDS);
    gu32 *status = (gu32 *)a.status;
    long long c0 = 0, c1 = 0, c2 = 0;
    if (ws.prof)
        c0 = clock64();
    // ---- (1) h_{step-1} of set M: tag == step, slot (step-1)&1
    if (step > 0)
    {
        gu64 *g = gran + a.set[M].gran_off + granule_index((step - 1) & 1, chain, w * KPW + l, a.S);
        const unsigned want = a.tag_base + (unsigned)step;
        unsigned long long x = WF_NO_PREFETCH ? 0ull : ws.pend[M];
        unsigned spins = 0;
        while (!__all((unsigned)(x >> 32) == want))
        {
            x = granule_load(g);
            if (++spins > LSTM_SPIN_LIMIT ||
                ((spins & 255u) == 0 && (*(volatile lds_i32 *)abort_flag ||
                                         __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            {
                if (l == 0)
                    __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(volatile lds_i32 *)abort_flag = 1;
                return false;
            }
        }
        ws.hval[M] = __uint_as_float((unsigned)x);
    }
    // ---- (2) issue the poll of the next task's set now: it completes behind this task's arithmetic.
    // Unconditional (a branch here makes hipcc drain vmcnt at the join): for nstep == 0 or T the address is
    // still inside the granule area and the value is never looked at.
    {
        constexpr int q = wf_next(MASK, M);
        const int nstep = step + wf_wrap(MASK, M);
        ws.pend[q] = granule_load(gran + a.set[q].gran_off + granule_index((nstep - 1) & 1, chain, w * KPW + l, a.S));
    }
    if (ws.prof)
        c1 = clock64();
    // ---- (3) MACs, fused with the pending gates of the previous set on its gate wave
    constexpr int G = WF_NO_FUSE ? M : wf_prev(MASK, M);
    if (G != M)
    {
        const int gstep = step - (G > M ? 1 : 0); // the last set of the round is finished at the start of the next step
        if (gate_duty && w == G && gstep >= 0)
        {
            if (!wf_wait_partials<G>(a, gstep, l, cnt, abort_flag))
                return false;
            if (ws.prof)
                c2 = clock64();
            wf_gates<G, FAST, PRECISE>(a, ws, gstep, G > M ? ws.p_old : ws.p, chain, target, dir, unit, l, part);
            wf_dot<M>(ws, step, w, l, part, cnt); // same basic block as the gates: the scheduler interleaves them
            if (ws.prof)
            {
                const long long c3 = clock64();
                ws.pc[0] += (unsigned long long)(c1 - c0);
                ws.pc[2] += (unsigned long long)(c2 - c1);
                ws.pc[3] += (unsigned long long)(c3 - c2);
                ws.pc[4] += 1;
            }
        }
        else
        {
            wf_dot<M>(ws, step, w, l, part, cnt);
            if (ws.prof)
            {
                const long long c3 = clock64();
                ws.pc[0] += (unsigned long long)(c1 - c0);
                ws.pc[1] += (unsigned long long)(c3 - c1);
                ws.pc[4] += 1;
            }
        }
    }
    else
    {
        wf_dot<M>(ws, step, w, l, part, cnt);
        if (gate_duty && w == M)
        {
            if (!wf_wait_partials<M>(a, step, l, cnt, abort_flag))
                return false;
            wf_gates<M, FAST, PRECISE>(a, ws, step, ws.p, chain, target, dir, unit, l, part);
        }
    }
    return true;
}

template <int MASK, bool FAST, bool PRECISE>
__device__ __forceinline__ void lstm_wavefront_body(const LstmWaveArgs &a, int chain, int slice,
                                                    float (*part)[2][8][64], lds_u32 *cnt, lds_i32 *abort_flag)
{
    constexpr int Hl = LSTM_WF_HL, KPW = LSTM_WF_KPW;
    const int target = a.tmap[chain >> 1], dir = chain & 1, wchain = target * 2 + dir;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int T = a.T;

    // ---- register-resident W_hh slices of the active sets (DPP layout, see lstm_kernels.h)
    WfWave ws;
    WSlice(&Wd)[3] = ws.Wd;
    {
        int kidx[16];
        KidxDpp<15>::run(w * KPW + l, kidx);
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if ((MASK >> m) & 1)
            {
                const float *Wb = a.set[m].W + ((size_t)wchain * a.S + slice) * Hl * 64 + 4 * (l & 15);
#pragma unroll
                for (int n = 0; n < 16; ++n)
                {
                    const float4 v = *reinterpret_cast<const float4 *>(Wb + (size_t)kidx[n] * 64);
                    Wd[m].set(n, v);
                }
            }
    }
    // ---- per-set state of this wave
    float(&hval)[3] = ws.hval;
    hval[0] = hval[1] = hval[2] = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
        if ((MASK >> m) & 1)
            hval[m] = a.state[state_off(target, a.set[m].layer, dir, 0, Hl) + w * KPW + l];
    // gate duty: wave m owns set m (c, b_hh, P column, output column)
    const int gm = w < 3 ? w : 0;
    const bool gate_duty = w < 3 && ((MASK >> gm) & 1);
    const int unit = slice * 16 + (l >> 2);
    float &c = ws.c, &bh = ws.bh, &hlast = ws.hlast;
    c = 0.f;
    bh = 0.f;
    hlast = 0.f;
    ws.p = ws.p_old = 0.f;
    const float *Pp = nullptr;
    if (gate_duty)
    {
        c = a.state[state_off(target, a.set[gm].layer, dir, 1, Hl) + unit];
        bh = a.set[gm].bhh[((size_t)wchain * a.S + slice) * 64 + l];
        Pp = a.set[gm].P[target] + ((size_t)dir * a.S + slice) * 64 + l;
    }
    ws.pend[0] = ws.pend[1] = ws.pend[2] = 0;
    ws.prof = a.prof != nullptr && chain == 0 && slice == 0 && l == 0 && (w == 0 || w == 3);
    for (int i = 0; i < 5; ++i)
        ws.pc[i] = 0;

    for (int step = 0; step < T; ++step)
    {
        const int t = dir == 0 ? step : T - 1 - step;
        ws.p_old = ws.p;
        ws.p = 0.f;
        if (gate_duty)
            ws.p = Pp[(size_t)t * a.ldp]; // one whole step ahead of its use
        if (!wf_task<MASK, 0, FAST, PRECISE>(a, ws, step, chain, target, dir, unit, gate_duty, part, cnt, abort_flag))
            return;
        if (WF_TASK_BARRIER && (MASK & 1))
            __syncthreads();
        if (!wf_task<MASK, 1, FAST, PRECISE>(a, ws, step, chain, target, dir, unit, gate_duty, part, cnt, abort_flag))
            return;
        if (WF_TASK_BARRIER && (MASK & 2))
            __syncthreads();
        if (!wf_task<MASK, 2, FAST, PRECISE>(a, ws, step, chain, target, dir, unit, gate_duty, part, cnt, abort_flag))
            return;
        if (WF_TASK_BARRIER && (MASK & 4))
            __syncthreads();
    }
    // the gates of the round's last set for step T-1 are still pending when more than one set is active
    if (!WF_NO_FUSE)
    {
        constexpr int LAST = (MASK & 4) ? 2 : (MASK & 2) ? 1 : 0;
        constexpr int FIRST = (MASK & 1) ? 0 : (MASK & 2) ? 1 : 2;
        if (LAST != FIRST && gate_duty && w == LAST)
        {
            if (!wf_wait_partials<LAST>(a, T - 1, l, cnt, abort_flag))
                return;
            wf_gates<LAST, FAST, PRECISE>(a, ws, T - 1, ws.p, chain, target, dir, unit, l, part);
        }
    }
    if (ws.prof)
        for (int i = 0; i < 5; ++i)
            a.prof[(w == 0 ? 0 : 1) * 8 + i] = ws.pc[i];
    if (gate_duty && (l & 3) == 0) // lstm.cpp:160-161: the state carries into the next segment
    {
        a.state[state_off(target, a.set[gm].layer, dir, 0, Hl) + unit] = hlast;
        a.state[state_off(target, a.set[gm].layer, dir, 1, Hl) + unit] = c;
    }
}

// grid = 8*S workgroups (1-D), plain launch (residency checked on the host), same census as
// lstm_persistent_kernel.
template <int MASK, bool PRECISE> __global__ __launch_bounds__(LSTM_THREADS) void lstm_wavefront_kernel(LstmWaveArgs a)
{
    __shared__ float part[3][2][8][64];
    __shared__ unsigned cnt[4];
    __shared__ int s_ctl[4]; // chain, slice, fast, abort
    const int tid = threadIdx.x;
    const int nwg = gridDim.x, S = a.S;
    if (tid < 4)
        cnt[tid] = 0;
    if (tid == 0)
    {
        gu32 *census = (gu32 *)a.sync;
        gu32 *arrived = (gu32 *)(a.sync + 8);
        gu32 *status = (gu32 *)a.status;
        const unsigned xcc = xcc_id() & 7;
        const unsigned ticket = __hip_atomic_fetch_add(census + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int abort_ = 0;
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg)
        {
            if (++spins > LSTM_SPIN_LIMIT)
            {
                __hip_atomic_store(status, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                abort_ = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        bool fast = !a.force_safe && !abort_;
        for (int x = 0; x < 8; ++x)
            fast = fast && __hip_atomic_load(census + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)S;
        s_ctl[0] = fast ? (int)xcc : (int)(blockIdx.x / S);
        s_ctl[1] = fast ? (int)ticket : (int)(blockIdx.x % S);
        s_ctl[2] = fast;
        s_ctl[3] = abort_;
        if (blockIdx.x == 0)
            __hip_atomic_store(status + 1, fast ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int chain = s_ctl[0], slice = s_ctl[1];
    if (s_ctl[3] || chain >= a.nchains)
        return;
    if (s_ctl[2])
        lstm_wavefront_body<MASK, true, PRECISE>(a, chain, slice, part, (lds_u32 *)cnt, (lds_i32 *)&s_ctl[3]);
    else
        lstm_wavefront_body<MASK, false, PRECISE>(a, chain, slice, part, (lds_u32 *)cnt, (lds_i32 *)&s_ctl[3]);
}

} // namespace umx
