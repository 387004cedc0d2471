// lstm_kernels.h -- the 3-layer bidirectional streaming LSTM recurrence (lstm.cpp:101-179).
//
// The input projection W_ih x_t + b_ih for all frames is a batched MFMA GEMM (gemm_bf16x3.h /
// gemm_kernels.h, mode G_IH) -- legal because it has no recurrence.  What is left per (target, layer, dir)
// "chain" is strictly serial:  gates_t = (P_t + W_hh h_{t-1}) + b_hh, i|f|g|o split
// (lstm.cpp:143-152), c = sig(f) c + sig(i) tanh(g), h = sig(o) tanh(c).
// 4 targets x 2 directions = 8 independent chains per layer run concurrently.
//
// Work split: one workgroup (512 threads, 8 waves) per (chain, slice of 16 hidden units).
// A slice owns the 64 gate columns of its units; lane l of every wave is gate column
// l = 4*u + g (unit u, gate g: the four gates of a unit sit in one DPP quad); wave w owns the
// k-range [w*Hl/8, (w+1)*Hl/8) of the W_hh . h contraction.
//   W    float [chains][S][Hl][64]     (k-major; or the file's u8 in the same layout + per-chain scale/offset)
//   bhh  float [chains][S][64]
//   P    float [Tp][2][S][64] per target (GEMM output, columns permuted to match)
// For Hl = 512: S = 32 slices -> 256 workgroups, W slice = 128 KiB fp32 = 64 VGPRs per lane.
//
// Two drivers share the same per-step arithmetic (bitwise-identical results):
//   lstm_step_kernel        one launch per timestep, h/c in HBM between launches (simple, safe)
//   lstm_persistent_kernel  one launch per layer: W_hh stays in VGPRs for all T steps and h is
//     exchanged between the chain's workgroups through 8-byte {tag = step, value} granules
//     ("the data is the flag": one naturally aligned 8-byte store per value, polled with
//     L1-bypassing loads, no fences).  Each wave polls exactly the Hl/8 (<= 64) granules of its
//     own k-range, one per lane.  Hl = 512: lane (j, r) owns unit j's 4 gate columns and the 16 k of row r;
//     the polled h is rotated inside each 16-lane row with DPP row_ror and rotation pairs feed
//     v_pk_fma_f32 (smaller Hl: lane -> SGPR-pair broadcast with v_readlane), so h never goes through
//     LDS.  The W_ih x + b_ih rows arrive through an LDS ring filled 16 rows at a time with
//     global_load_lds, 16-32 steps ahead (a per-step HBM load in a polling wave would put the loaded HBM
//     latency in front of every step: vmcnt retires in order).  Two granule slots (step parity) are enough: a
//     producer can only overwrite slot p two steps later, which needs every consumer's next h,
//     i.e. every consumer is past its reads.
//     Placement: measured on MI355X (tools/handoff_probe.hip) a write-through (sc1) store +
//     sc1 load hand-off costs ~0.42-0.62 us one way and works for any placement; a plain store
//     + sc1 load costs ~0.25 us but is only coherent inside one XCD (one L2).  The kernel
//     therefore takes a CENSUS first: every workgroup registers on the XCD it actually runs on
//     (HW_REG_XCC_ID), one grid barrier, and only if every XCD received exactly S workgroups
//     does chain c become "the S workgroups on XCD c" with the fast intra-L2 protocol;
//     otherwise roles are static and the protocol is the placement-independent sc1 one.
//     Correctness never depends on where the dispatcher put a workgroup -- only speed does.
// Floor for one segment: 3 layers x 2584 steps = 7752 serially dependent steps.
#pragma once
#include "common.h"

namespace umx
{

typedef float float2v __attribute__((ext_vector_type(2)));
// .x + .y as ONE scalar v_add_f32.  Left to the compiler this becomes v_pk_add_f32 v, v, v op_sel:[0,1]
// op_sel_hi:[1,0]; packed fp32 ops whose low half selects the HIGH half of src1 return wrong results on MI355X
// while a co-resident wave issues v_mfma_f32_32x32x16_bf16 (tools/pk_mfma_probe.hip).  The same-register
// form measured clean, but this kernel shares its CUs with the bf16 GEMMs all the time: do not depend on it.
__device__ __forceinline__ float hadd2(float2v v)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
    return r;
}

struct LstmArgs
{
    const float *W;   // this layer: [chains][S][Hl][64]
    const unsigned char *Wq; // or (W == nullptr) the same layout as stored in the ggml file, u8: w = q*wsc+wof
    float wsc[8], wof[8];    // per chain (weight_hh_l{layer}[_reverse] of each target), model.cpp:610-616
    const float *bhh; // [chains][S][64]
    const float *P[4];
    float *out[4];    // out[target][t*ldo + col0 + dir*Hl + unit]
    float *state;     // [4 targets][3 layers][2 dirs][2 (h,c)][Hl]   (lstm.hpp:10-16 h, c)
    float *hbuf;      // step driver: [2][chains][Hl] ping-pong h
    unsigned *sync;   // persistent driver: [0..7] census per XCD, [8] arrivals, [16..] granules as u64 [2][8][Hl]
    unsigned *status; // [0] = abort/timeout flag (0 = ok), [1] = 1 if the fast intra-XCD protocol ran
    unsigned long long *prof; // optional phase-cycle counters (nullptr = off)
    int Hl, S, T, ldp, ldo, col0, layer, nchains;
    int tmap[4];      // chain>>1 -> target (targets can be skipped: BASELINE config 1)
    int force_safe;   // 1 = never use the intra-XCD protocol (testing)
    int poll_delay;    // x64 shader cycles a dot wave sleeps behind the barrier before its first poll (0 = LSTM_POLL_DELAY)
    int abort_at;      // testing (UMX_FLAG_DEBUG_LSTM_ABORT): every workgroup gives up at this step as if a poll had timed out
    unsigned tag_base; // granule tag of step s = tag_base + s + 1: unique per launch, so a granule line left in
                       // some L2 by an earlier launch can never pass for this launch's data
};

constexpr int LSTM_SYNC_HEADER_WORDS = 32; // census[8], arrivals, pad -> granules start 128-byte aligned
#define LSTM_SLICE_STRIDE 16 // granules (8 B each) between the 16-granule lines of consecutive slices
// granule index of (slot, chain, hidden unit k): one 128-byte line per producer slice, lines spread with
// LSTM_SLICE_STRIDE so that a chain's lines do not pile up on one L2 channel
__host__ __device__ inline size_t granule_index(int slot, int chain, int k, int S)
{
    return ((size_t)(slot * 8 + chain) * S + (k >> 4)) * LSTM_SLICE_STRIDE + (k & 15);
}
__host__ __device__ inline size_t granule_count(int S) { return (size_t)2 * 8 * S * LSTM_SLICE_STRIDE; }

// element (k, col) of a chain-slice's W_hh block [Hl][64], fp32-resident or u8-resident
__device__ __forceinline__ float whh_at(const LstmArgs &a, int wchain, size_t idx)
{
    return a.W ? a.W[idx] : (float)a.Wq[idx] * a.wsc[wchain] + a.wof[wchain];
}
__device__ __forceinline__ float whh_at(const float *W, const unsigned char *Wq, float sc, float of, size_t idx)
{
    return W ? W[idx] : (float)Wq[idx] * sc + of;
}
__device__ __forceinline__ float4 whh_at4(const LstmArgs &a, int wchain, size_t idx) // idx % 4 == 0
{
    if (a.W)
        return *reinterpret_cast<const float4 *>(a.W + idx);
    const unsigned p = *reinterpret_cast<const unsigned *>(a.Wq + idx);
    const float sc = a.wsc[wchain], of = a.wof[wchain];
    return make_float4((float)(p & 255u) * sc + of, (float)((p >> 8) & 255u) * sc + of,
                       (float)((p >> 16) & 255u) * sc + of, (float)(p >> 24) * sc + of);
}

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); } // lstm.cpp:36-39

// Activations.  PRECISE = the ROCm device-library expf/tanhf (<= 2 ulp) with IEEE division; the default
// uses the hardware transcendentals directly (v_exp_f32, v_rcp_f32; ~1e-7 absolute error), which is the
// same error class as the reference's own Eigen vectorised exp/tanh (generic_fast_tanh_float) and
// shortens the serial gate phase of every LSTM step by ~500 shader cycles.
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
// tanh from e = exp(-2|x|): (1 - e) / (1 + e) with the sign of x; |x| < 0.125 uses the odd Taylor polynomial
// through x^7 (next term < 2e-10 relative) instead, which avoids the 1 - e cancellation
__device__ __forceinline__ float tanh_from_e(float x, float e, float rcp1pe)
{
    const float x2 = x * x;
    const float big = copysignf((1.0f - e) * rcp1pe, x);
    const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -0.0539682540f, 0.133333333f), -0.333333333f), 1.0f);
    return fabsf(x) < 0.125f ? poly : big;
}
__device__ __forceinline__ float tanh_hw(float x)
{
    const float e = exp_hw(-2.0f * fabsf(x));
    return tanh_from_e(x, e, __builtin_amdgcn_rcpf(1.0f + e));
}

// pre-activation of gate column lane = 4*u + g (all 64 lanes) -> new (c, h), replicated in the quad
template <bool PRECISE> __device__ __forceinline__ void lstm_cell(float pre, int lane, float &c, float &h)
{
    const int g = lane & 3;
    float act;
    if (PRECISE)
        act = (g == 2) ? tanhf(pre) : sigmoid_ref(pre);
    else
    {
        // one v_exp_f32 + one v_rcp_f32 per lane, no divergence: e = exp(-x) (sigmoid lanes) or exp(-2|x|)
        const float e = exp_hw(g == 2 ? -2.0f * fabsf(pre) : -pre);
        const float r = __builtin_amdgcn_rcpf(1.0f + e);
        act = (g == 2) ? tanh_from_e(pre, e, r) : r;
    }
    const int ai = __float_as_int(act);
    const float i_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0x00, 0xF, 0xF, true)); // quad_perm [0,0,0,0]
    const float f_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0x55, 0xF, 0xF, true)); // [1,1,1,1]
    const float g_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0xAA, 0xF, 0xF, true)); // [2,2,2,2]
    const float o_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0xFF, 0xF, 0xF, true)); // [3,3,3,3]
    const float c_t = f_t * c + i_t * g_t; // lstm.cpp:154-156
    c = c_t;
    h = o_t * (PRECISE ? tanhf(c_t) : tanh_hw(c_t)); // lstm.cpp:157
}

// The same arithmetic as lstm_cell<false>, operation for operation (same bits), written without control flow: the
// gate wave's phase is ONE dependent chain on the critical path of every step, and the two exec-mask branches the
// compiler makes of lstm_cell (tanh only in the g lanes, tanh(c) only in the storing lanes) cost more than the few
// instructions they skip.  Per-lane constants (g lane or not) are loop-invariant registers of the caller.
struct CellLane
{
    unsigned abs_mask; // g lanes: 0x7fffffff (|pre|), others: all ones
    float zk;          // g lanes: -2, others: -1        (exp argument: -2|x| or -x)
    unsigned tanh_sel; // g lanes: all ones (take tanh), others 0 (take the sigmoid)
    __device__ __forceinline__ void init(int lane)
    {
        const bool g = (lane & 3) == 2;
        abs_mask = g ? 0x7fffffffu : 0xffffffffu;
        zk = g ? -2.0f : -1.0f;
        tanh_sel = g ? 0xffffffffu : 0u;
    }
};
__device__ __forceinline__ float tanh_from_e_sel(float x, float e, float rcp1pe)
{
    const float x2 = x * x;
    const float big = copysignf((1.0f - e) * rcp1pe, x);
    const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -0.0539682540f, 0.133333333f), -0.333333333f), 1.0f);
    const bool small = fabsf(x) < 0.125f;
    return small ? poly : big; // both sides are values already: a v_cndmask, no branch
}
__device__ __forceinline__ void lstm_cell_flat(float pre, const CellLane &cl, float &c, float &h)
{
    const float xa = __uint_as_float(__float_as_uint(pre) & cl.abs_mask);
    const float e = exp_hw(cl.zk * xa); // exp(-2|x|) in the g lanes, exp(-x) elsewhere: lstm_cell's argument bit for bit
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    const float th = tanh_from_e_sel(pre, e, r);
    const int ai = (int)((__float_as_uint(th) & cl.tanh_sel) | (__float_as_uint(r) & ~cl.tanh_sel)); // v_bfi_b32
    const float i_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0x00, 0xF, 0xF, true));
    const float f_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0x55, 0xF, 0xF, true));
    const float g_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0xAA, 0xF, 0xF, true));
    const float o_t = __int_as_float(__builtin_amdgcn_update_dpp(0, ai, 0xFF, 0xF, 0xF, true));
    const float c_t = f_t * c + i_t * g_t; // lstm.cpp:154-156
    c = c_t;
    const float e2 = exp_hw(-2.0f * fabsf(c_t));
    h = o_t * tanh_from_e_sel(c_t, e2, __builtin_amdgcn_rcpf(1.0f + e2)); // lstm.cpp:157
}

// row_ror:n of a 32-bit value (rotation inside each row of 16 lanes); n is a compile-time constant
template <int N> __device__ __forceinline__ int dpp_row_ror(int v)
{
    if (N == 0)
        return v;
    return __builtin_amdgcn_mov_dpp(v, 0x120 + (N & 15), 0xF, 0xF, true);
}
// +1 if row_ror:1 makes lane i read lane i+1, -1 if it reads lane i-1 (self-calibrating: the step
// driver must walk k in the same order as the persistent kernel's DPP rotations)
__device__ __forceinline__ int dpp_ror_direction()
{
    const int lane = threadIdx.x & 63;
    const int src = dpp_row_ror<1>(lane);
    return (((src - lane) & 15) == 1) ? 1 : -1;
}

__host__ __device__ __forceinline__ size_t state_off(int target, int layer, int dir, int hc, int Hl)
{
    return ((((size_t)target * 3 + layer) * 2 + dir) * 2 + hc) * Hl;
}

// ------------------------------------------------------------------ per-step driver
// grid (S, chains); hbuf[step&1] holds h_{t-1}, hbuf[(step+1)&1] receives h_t; c lives in state.
// The k summation order mirrors the persistent kernel exactly (bitwise-identical results):
//   Hl == 512: per wave four row-partials p_r over k = 64w + 16r + ((u + dir*n) & 15), n = 0..15
//              (the DPP rotation order), combined as (p0 + p2) + (p1 + p3);
//   otherwise: even-k / odd-k partial sums (the v_pk_fma_f32 order).
template <bool PRECISE> __global__ __launch_bounds__(LSTM_THREADS) void lstm_step_kernel(LstmArgs a, int step)
{
    __shared__ float hs[1024];
    __shared__ float part[8][64];
    const int slice = blockIdx.x, chain = blockIdx.y, target = a.tmap[chain >> 1], dir = chain & 1;
    const int wchain = target * 2 + dir; // weights are stored for all 8 chains
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int Hl = a.Hl, nchains = gridDim.y;
    const int t = dir == 0 ? step : a.T - 1 - step; // lstm.cpp:118-120
    const float *hprev = a.hbuf + ((size_t)(step & 1) * nchains + chain) * Hl;
    for (int i = tid; i < Hl; i += LSTM_THREADS)
        hs[i] = hprev[i];
    __syncthreads();
    const int kpw = Hl >> 3;
    const size_t Wc0 = ((size_t)wchain * a.S + slice) * Hl * 64 + l; // column l, stride 64 per k
    float partial;
    if (kpw == 64)
    {
        const int rdir = dpp_ror_direction(), u = l >> 2;
        float pr[4];
        for (int r = 0; r < 4; ++r)
        {
            float acc_e = 0.f, acc_o = 0.f; // even / odd rotations: the two halves of v_pk_fma_f32
            for (int n = 0; n < 16; n += 2)
            {
                const int ke = 64 * w + 16 * r + ((u + rdir * n) & 15);
                const int ko = 64 * w + 16 * r + ((u + rdir * (n + 1)) & 15);
                acc_e = fmaf(whh_at(a, wchain, Wc0 + (size_t)ke * 64), hs[ke], acc_e);
                acc_o = fmaf(whh_at(a, wchain, Wc0 + (size_t)ko * 64), hs[ko], acc_o);
            }
            pr[r] = acc_e + acc_o;
        }
        partial = (pr[0] + pr[2]) + (pr[1] + pr[3]);
    }
    else
    {
        float2v acc = {0.f, 0.f};
        for (int i = 0; i < kpw; i += 2)
        {
            float2v wv, hv;
            wv.x = whh_at(a, wchain, Wc0 + (size_t)(w * kpw + i) * 64);
            wv.y = whh_at(a, wchain, Wc0 + (size_t)(w * kpw + i + 1) * 64);
            hv.x = hs[w * kpw + i];
            hv.y = hs[w * kpw + i + 1];
            acc = __builtin_elementwise_fma(wv, hv, acc);
        }
        partial = hadd2(acc);
    }
    part[w][l] = partial;
    __syncthreads();
    if (w == 0)
    {
        const float s = ((part[0][l] + part[1][l]) + (part[2][l] + part[3][l])) +
                        ((part[4][l] + part[5][l]) + (part[6][l] + part[7][l]));
        const float p = a.P[target][(size_t)t * a.ldp + ((size_t)dir * a.S + slice) * 64 + l];
        const float pre = (p + s) + a.bhh[((size_t)wchain * a.S + slice) * 64 + l]; // lstm.cpp:132-140
        const int unit = slice * 16 + (l >> 2);
        float *cst = a.state + state_off(target, a.layer, dir, 1, Hl);
        float c = cst[unit], h;
        lstm_cell<PRECISE>(pre, l, c, h);
        if ((l & 3) == 0)
        {
            cst[unit] = c;
            a.hbuf[((size_t)((step + 1) & 1) * nchains + chain) * Hl + unit] = h;
            a.out[target][(size_t)t * a.ldo + a.col0 + dir * Hl + unit] = h; // lstm.cpp:163-164,170-171
        }
    }
}

// copies between the stream state and the ping-pong buffer around a layer of step launches
__global__ void lstm_state_to_hbuf(LstmArgs a, int nchains)
{
    const int chain = blockIdx.x, target = a.tmap[chain >> 1], dir = chain & 1;
    for (int i = threadIdx.x; i < a.Hl; i += blockDim.x)
        a.hbuf[(size_t)chain * a.Hl + i] = a.state[state_off(target, a.layer, dir, 0, a.Hl) + i];
}
__global__ void lstm_hbuf_to_state(LstmArgs a, int nchains)
{
    const int chain = blockIdx.x, target = a.tmap[chain >> 1], dir = chain & 1;
    const float *src = a.hbuf + ((size_t)(a.T & 1) * nchains + chain) * a.Hl;
    for (int i = threadIdx.x; i < a.Hl; i += blockDim.x)
        a.state[state_off(target, a.layer, dir, 0, a.Hl) + i] = src[i];
}

// ------------------------------------------------------------------ persistent driver
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr unsigned LSTM_SPIN_LIMIT = 1u << 22; // bounded spins: ~seconds, then abort the launch
#define LSTM_POLLS_IN_FLIGHT 1 // measured best alone and with two grids on the chip (round 5, alone: 2: +3 %, 3: +10 %; round 1, pipelined: 2: +3 %, 3: +5 %)
#define LSTM_TRACE_SLICE 5
#define LSTM_TRACE_STEP0 1200
#define LSTM_PROF_WAVE 1 // the dot wave the in-kernel profiler reports beside the gate wave
#define LSTM_P_BULK 16 // W_ih x + b_ih rows fetched per bulk (multiple of 8, power of two)
#define LSTM_P_RING (2 * LSTM_P_BULK)
#define LSTM_GATE_POLL_DELAY 0 // x64 cycles the gate wave waits after publishing before its own first poll
                               // (measured 0..4: 0 is best, its first poll already succeeds)
#define LSTM_POLL_DELAY 5 // x64 shader cycles a dot wave sleeps after the barrier before its first poll (round 5: 8 -> 5 with the shorter gate phase; 6: +2 %, 4: +2 %, 3: +5 %)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xF;
}

// 8-byte granule store / load.  FAST (all parties share one XCD L2): plain store, L1-bypassing load.
// SAFE (any placement): write-through agent-scope store + agent-scope load.
template <bool FAST> __device__ __forceinline__ void granule_store(gu64 *p, unsigned long long v)
{
    if (FAST)
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ... sc1
}
__device__ __forceinline__ unsigned long long granule_load(gu64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_load_dwordx2 ... sc1
}

// Census (thread 0 of every workgroup): register on the XCD this workgroup actually runs on, one bounded grid
// barrier, then roles: if every XCD received exactly S workgroups, chain = XCD and slice = arrival ticket with the
// intra-L2 protocol (ctl[2] = 1); otherwise static roles and the placement-independent sc1 protocol.
// ctl = {chain, slice, fast, abort}
__device__ __forceinline__ void lstm_census(unsigned *sync, unsigned *status_, int S, int nwg, int force_safe, int *ctl)
{
    gu32 *census = (gu32 *)sync;
    gu32 *arrived = (gu32 *)(sync + 8);
    gu32 *status = (gu32 *)status_;
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
    bool fast = !force_safe && !abort_;
    for (int x = 0; x < 8; ++x)
        fast = fast && __hip_atomic_load(census + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)S;
    ctl[0] = fast ? (int)xcc : (int)(blockIdx.x / S);
    ctl[1] = fast ? (int)ticket : (int)(blockIdx.x % S);
    ctl[2] = fast;
    ctl[3] = abort_;
    if (blockIdx.x == 0)
        __hip_atomic_store(status + 1, fast ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// KPW = Hl/8 = k-range (and granules) per wave; KPW <= 64, even.
// register-resident W_hh slice of one lane: rotation pairs (2m, 2m+1) are adjacent registers
struct WSlice
{
    float2v v[8][4];
    __device__ __forceinline__ void set(int n, const float4 &x)
    {
        v[n >> 1][0][n & 1] = x.x;
        v[n >> 1][1][n & 1] = x.y;
        v[n >> 1][2][n & 1] = x.z;
        v[n >> 1][3][n & 1] = x.w;
    }
};
// acc2[cc] += (W[2M][cc], W[2M+1][cc]) * (h ror 2M, h ror 2M+1) for M = 0 .. 7; the caller adds .x + .y
template <int M> struct DotDppPk
{
    static __device__ __forceinline__ void run(const WSlice &W, int hbits, float2v (&acc2)[4])
    {
        DotDppPk<M - 1>::run(W, hbits, acc2);
        float2v hr;
        hr.x = __int_as_float(dpp_row_ror<2 * M>(hbits));
        hr.y = __int_as_float(dpp_row_ror<2 * M + 1>(hbits));
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            acc2[cc] = __builtin_elementwise_fma(W.v[M][cc], hr, acc2[cc]);
    }
};
template <> struct DotDppPk<-1>
{
    static __device__ __forceinline__ void run(const WSlice &, int, float2v (&)[4]) {}
};
template <int N> struct DotDpp
{
    static_assert(N == 15, "full 16-rotation dot only");
    static __device__ __forceinline__ void run(const WSlice &W, int hbits, float (&acc)[4])
    {
        float2v acc2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        DotDppPk<7>::run(W, hbits, acc2);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            acc[cc] = hadd2(acc2[cc]);
    }
};
template <int N> struct KidxDpp
{
    static __device__ __forceinline__ void run(int lane_k, int (&kidx)[16])
    {
        KidxDpp<N - 1>::run(lane_k, kidx);
        kidx[N] = dpp_row_ror<N>(lane_k);
    }
};
template <> struct KidxDpp<-1>
{
    static __device__ __forceinline__ void run(int, int (&)[16]) {}
};

template <int KPW, bool FAST, bool PRECISE>
__device__ __forceinline__ void lstm_persistent_body(const LstmArgs &a, int chain, int slice, float (*part)[8][64],
                                                      float (*pbuf)[64], int *abort_flag)
{
    const int target = a.tmap[chain >> 1], dir = chain & 1;
    const int wchain = target * 2 + dir;
    // w (and with it every role test) is wave-uniform: keep it in an SGPR so role branches are scalar and
    // the profiler's counters cost no VGPRs
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    constexpr int Hl = KPW * 8;
    const int T = a.T;

    // W_hh slice -> registers, once.
    // KPW == 64 ("DPP" layout): lane (j = l & 15, r = l >> 4) owns the 4 gate columns of unit j and the
    //   16 k of row r: Wd[n][cc] = W[k_n][4j + cc] where k_n is the k whose h this lane sees after
    //   row_ror:n of the polled h vector (self-calibrated with the same DPP op) -> 64 v_fmac_f32_dpp
    //   per step, no lane->SGPR broadcast at all.
    // otherwise ("readlane" layout): lane l owns gate column l and all KPW k of the wave: pairs
    //   (k even, k odd) feed v_pk_fma_f32 with an SGPR pair from v_readlane.
    constexpr bool DPP = (KPW == 64);
    float2v W[KPW / 2]; // dead in the DPP instantiation
    WSlice Wd;          // dead in the readlane instantiations
    if (w >= 8)
    {
        // gate wave: no weights
    }
    else if constexpr (DPP)
    {
        int kidx[16];
        KidxDpp<15>::run(w * KPW + l, kidx);
        const size_t Wb = ((size_t)wchain * a.S + slice) * Hl * 64 + 4 * (l & 15);
#pragma unroll
        for (int n = 0; n < 16; ++n)
        {
            const float4 v = whh_at4(a, wchain, Wb + (size_t)kidx[n] * 64);
            Wd.set(n, v);
        }
    }
    else
    {
        const size_t Wp = (((size_t)wchain * a.S + slice) * Hl + (size_t)w * KPW) * 64 + l;
#pragma unroll
        for (int i = 0; i < KPW / 2; ++i)
        {
            W[i].x = whh_at(a, wchain, Wp + (size_t)(2 * i) * 64);
            W[i].y = whh_at(a, wchain, Wp + (size_t)(2 * i + 1) * 64);
        }
    }
    // Wave roles: waves 0..7 poll h, multiply their k-range and leave 64 partial sums in LDS; wave 8
    // (the "gate wave") owns c, adds the 8 partials, applies the gates and publishes the 16 new h.
    // One barrier per step.  Keeping the gates out of the dot waves matters: a wave that computed the
    // gates would start polling ~650 cycles after the others and miss a whole L2 round trip every step.
    // blockDim 576: dedicated gate wave 8; blockDim 512: wave 0 also applies the gates (two such
    // workgroups fit one CU, which the cross-segment pipeline of engine.hip relies on)
    const int gw = (blockDim.x > 512) ? 8 : 0;
    const bool gate_wave = (w == gw), dot_wave = (w < 8);
    const float bh = gate_wave ? a.bhh[((size_t)wchain * a.S + slice) * 64 + l] : 0.f;
    const int unit = slice * 16 + (l >> 2);
    float c = 0.f;
    if (gate_wave)
        c = a.state[state_off(target, a.layer, dir, 1, Hl) + unit];
    // h_{-1}: this wave's k-range, one value per lane (lanes >= KPW idle)
    float hval = 0.f;
    if (dot_wave && l < KPW)
        hval = a.state[state_off(target, a.layer, dir, 0, Hl) + w * KPW + l];

    gu64 *gran = (gu64 *)(a.sync + LSTM_SYNC_HEADER_WORDS);
    gu32 *status = (gu32 *)a.status;
    // everything indexed by the runtime `target` is resolved once, outside the step loop (a dynamic index
    // into the by-value argument struct is a memory load, and the granule store's compiler barrier would
    // otherwise force it back into every step, right on the gate wave's critical path)
    const float *const Pp = a.P[target] + ((size_t)dir * a.S + slice) * 64 + l;
    float *const outp = a.out[target] + a.col0 + dir * Hl + unit;
    const size_t ldp = (size_t)a.ldp, ldo = (size_t)a.ldo;
    const unsigned tag_base = a.tag_base;
    const int S = a.S;
    const int poll_delay = a.poll_delay > 0 ? a.poll_delay : LSTM_POLL_DELAY;
    float hlast = 0.f;
    // Gate wave, loop-invariant: the per-lane constants of the branch-free cell, and the two stores of a step as buffer
    // stores whose offset is out of range in the lanes that do not store (one lane per unit does): no exec-mask branch
    // and no 64-bit address arithmetic on the critical path between h and its publication.
    CellLane cl;
    cl.init(l);
    const bool store_lane = (l & 3) == 0;
    const __amdgpu_buffer_rsrc_t gran_rs = __builtin_amdgcn_make_buffer_rsrc(a.sync + LSTM_SYNC_HEADER_WORDS, 0, (int)(granule_count(S) * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(outp - unit, 0, 0x7fffffff, 0x00020000);
    const unsigned goff[2] = {store_lane ? (unsigned)(granule_index(0, chain, unit, S) * 8) : 0x80000000u,
                              store_lane ? (unsigned)(granule_index(1, chain, unit, S) * 8) : 0x80000000u};
    const unsigned ooff = store_lane ? (unsigned)unit * 4u : 0x80000000u;
    // W_ih x + b_ih rows come from HBM, and vmcnt retires in order: whichever wave has such a load in flight
    // cannot complete its next hidden-state poll before the row has arrived.  Alone on the chip that is ~750
    // cycles; beside a GEMM or a streaming kernel of the other pipeline slot it is several microseconds, and a
    // row fetched every step put exactly that in front of every step (an LSTM launch sharing the chip with a
    // copy kernel ran 3x slower).  So rows are fetched in bulk, LSTM_P_BULK at a time and LSTM_P_BULK..2x steps
    // ahead, straight into an LDS ring (global_load_lds: no VGPRs, no ds_write), by waves 1..7 at the top of a step: the
    // loaded-latency is paid once per LSTM_P_BULK steps instead of once per step.
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    const unsigned pbuf_lds = (unsigned)(size_t)(lds_ptr)&pbuf[0][0]; // LDS byte address of the ring
    // Waves 1..7 fetch (row j of a bulk: wave 1 + j % 7), at the TOP of the step that frees the slots and in front of their
    // sleep: the rows have the sleep and the poll's round trip (~900 cycles) to land before anything of that wave waits on
    // vmcnt.  Wave 0 fetches nothing: in the 512-thread form it is the gate wave, and its queue stays free for the gate
    // phase's stores and its own poll.  (Round 1-4 issued the fetch behind the successful poll, i.e. in front of the
    // multiply: any vmcnt wait the compiler places there -- it does -- then stalls the wave for the rows' HBM latency
    // once per bulk, 3-4 % of a launch.)
    auto fetch_rows = [&](int first_row) {
#pragma unroll
        for (int j = 0; j < LSTM_P_BULK; ++j)
        {
            const int r = first_row + j;
            if (w == 1 + j % 7 && r < T)
                __builtin_amdgcn_global_load_lds((glb_ptr)(Pp + (size_t)(dir == 0 ? r : T - 1 - r) * ldp),
                                                 (lds_ptr)(size_t)(pbuf_lds + 256u * (unsigned)(r & (LSTM_P_RING - 1))), 4, 0, 0);
        }
    };
    if (dot_wave)
    {
        fetch_rows(0);
        fetch_rows(LSTM_P_BULK);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): nothing loaded above (bias, state, the first rows) is still in flight inside the loop
    __syncthreads(); // the gate wave reads row 0 of the ring in front of the loop's first barrier: every wave's rows are in LDS
    const bool prof = a.prof != nullptr && chain == 0 && slice == 0 && (w == gw || w == LSTM_PROF_WAVE); // wave-uniform
    // timeline of the profiler: every wave of (chain 0, slice LSTM_TRACE_SLICE) stamps steps [LSTM_TRACE_STEP0, + 64)
    const bool trace = a.prof != nullptr && chain == 0 && slice == LSTM_TRACE_SLICE;
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
    unsigned prof_spins = 0;

    for (int step = 0; step < T; ++step)
    {
        const int t = dir == 0 ? step : T - 1 - step;
        long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        const bool stamp = prof || (trace && step >= LSTM_TRACE_STEP0 && step < LSTM_TRACE_STEP0 + 64);
        if (stamp)
            c0 = clock64();
        if (a.abort_at && step == a.abort_at && tid == 0)
        {
            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *abort_flag = 1;
        }
        if (dot_wave)
        {
            // rows [step + BULK, step + 2 BULK) replace rows [step - BULK, step), all consumed: this workgroup's gate wave read
            // row step - 1 before the barrier that ended step - 1.  They land in LDS before this wave's poll completes
            // (vmcnt in order), i.e. before it enters the next barrier, and are first read BULK barriers later.
            if (step > 0 && (step & (LSTM_P_BULK - 1)) == 0)
                fetch_rows(step + LSTM_P_BULK);
            if (step > 0)
            {
                // wait for h_{step-1}: tag == step, slot (step-1)&1.  A granule keeps its tag until it is
                // overwritten two steps later, so every lane reloads until one poll shows all tags.
                gu64 *g = gran + granule_index((step - 1) & 1, chain, w * KPW + (l < KPW ? l : 0), S);
                const unsigned want = tag_base + (unsigned)step;
                // The gate wave needs ~600 cycles before anything can change: sleep through that, then keep
                // three polls in flight so the poll period is a third of the L2 round trip (the step time is
                // a maximum over ~2000 polling waves: the quantisation is paid almost in full every step).
                if (FAST)
                {
                    if (!gate_wave)
                        for (int i = 0; i < poll_delay; ++i)
                            __builtin_amdgcn_s_sleep(1);
                    else if (LSTM_GATE_POLL_DELAY > 0)
                        __builtin_amdgcn_s_sleep(LSTM_GATE_POLL_DELAY);
                }
                unsigned long long x = granule_load(g), xb = 0, xc = 0;
                if (LSTM_POLLS_IN_FLIGHT >= 2)
                    xb = granule_load(g);
                if (LSTM_POLLS_IN_FLIGHT >= 3)
                    xc = granule_load(g);
                unsigned spins = 0;
                for (;;)
                {
                    if (__all((unsigned)(x >> 32) == want))
                        break;
                    if (LSTM_POLLS_IN_FLIGHT >= 3)
                    {
                        x = xb;
                        xb = xc;
                        xc = granule_load(g);
                    }
                    else if (LSTM_POLLS_IN_FLIGHT == 2)
                    {
                        x = xb;
                        xb = granule_load(g);
                    }
                    else
                        x = granule_load(g);
                    if (++spins > LSTM_SPIN_LIMIT ||
                        ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                    {
                        if (l == 0)
                            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *abort_flag = 1;
                        break;
                    }
                    if (!FAST)
                        __builtin_amdgcn_s_sleep(1);
                }
                prof_spins = spins;
                hval = __uint_as_float((unsigned)x);
            }
            if (stamp)
                c1 = clock64();
            if constexpr (DPP)
            {
                // 16 in-row rotations x 4 columns, 4 independent accumulators
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                DotDpp<15>::run(Wd, __float_as_int(hval), acc);
                // rows r and r+2 meet through v_permlane32_swap (lanes l <-> l+32): lane rows {0,1} then hold gates 0 and 2, rows {2,3} gates 1 and 3
                float s0, s1;
                {
                    const auto r01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0]), __float_as_uint(acc[1]), false, false);
                    const auto r23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[2]), __float_as_uint(acc[3]), false, false);
                    s0 = __uint_as_float(r01[0]) + __uint_as_float(r01[1]); // p_r + p_{r+2}
                    s1 = __uint_as_float(r23[0]) + __uint_as_float(r23[1]);
                }
                // rows r and r ^ 1 meet through ONE v_permlane16_swap of (s0, s1): afterwards row 0 holds both halves of
                // gate 0, row 1 of gate 2, row 2 of gate 1, row 3 of gate 3 -- every lane ends with one finished column
                // (the same (p0 + p2) + (p1 + p3) as before, without the LDS round trip of two ds_swizzle)
                const auto rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(s0), __float_as_uint(s1), false, false);
                const float tsum = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
                (*(part + (step & 1)))[w][4 * (l & 15) + (((l >> 4) & 1) << 1) + (l >> 5)] = tsum;
            }
            else
            {
                // partial dot product over this wave's k-range, h broadcast lane -> SGPR pair
                float2v acc = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KPW / 2; ++i)
                {
                    float2v hk;
                    hk.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hval), 2 * i));
                    hk.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hval), 2 * i + 1));
                    acc = __builtin_elementwise_fma(W[i], hk, acc);
                }
                (*(part + (step & 1)))[w][l] = hadd2(acc);
            }
        }
        // the gate wave's W_ih x + b_ih element of this step has been in the ring for >= LSTM_P_BULK steps: read it before the
        // barrier, so that behind the barrier only the eight partial sums stand between the wave and the activations
        float prow = 0.f;
        if (gate_wave)
            prow = pbuf[step & (LSTM_P_RING - 1)][l];
        if (stamp)
            c2 = clock64();
        // workgroup barrier for LDS only (the partial sums of this step): __syncthreads() is also a release fence for global
        // memory, i.e. s_waitcnt vmcnt(0) in front of the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (*abort_flag) // uniform after the barrier: every wave leaves, nobody is left spinning on us
            return;
        if (stamp)
            c3 = clock64();
        if (gate_wave)
        {
            __builtin_amdgcn_s_setprio(1); // the serial gate phase wins issue arbitration against dot waves
            float(*pq)[64] = *(part + (step & 1));
            const float s = ((pq[0][l] + pq[1][l]) + (pq[2][l] + pq[3][l])) + ((pq[4][l] + pq[5][l]) + (pq[6][l] + pq[7][l]));
            const float pre = (prow + s) + bh; // lstm.cpp:132-140; prow was read from the ring before the barrier
            float h;
            if constexpr (PRECISE)
                lstm_cell<true>(pre, l, c, h);
            else
                lstm_cell_flat(pre, cl, c, h);
            typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
            v2u32 gv;
            gv.x = __float_as_uint(h);
            gv.y = tag_base + (unsigned)(step + 1);
            // "the data is the flag": one naturally aligned 8-byte store {value, tag}; FAST = plain (the chain shares one L2), SAFE = sc1
            __builtin_amdgcn_raw_buffer_store_b64(gv, gran_rs, goff[step & 1], 0, FAST ? 0 : 16);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), out_rs, ooff, (unsigned)t * (unsigned)ldo * 4u, 0); // lstm.cpp:163-164
            hlast = h;
            __builtin_amdgcn_s_setprio(0);
        }
        if (stamp && !prof)
        {
            const long long c4 = clock64();
            if (l == 0)
            {
                unsigned long long *tr = a.prof + 1024 + ((size_t)(step - LSTM_TRACE_STEP0) * 8 + (w & 7)) * 5;
                tr[0] = c0, tr[1] = c1, tr[2] = c2, tr[3] = c3, tr[4] = c4;
            }
        }
        if (prof)
        {
            const long long c4 = clock64();
            pc[0] += (unsigned long long)(c1 - c0); // poll (dot waves)
            pc[1] += (unsigned long long)(c2 - c1); // dot
            pc[2] += (unsigned long long)(c3 - c2); // barrier wait
            pc[3] += (unsigned long long)(c4 - c3); // gates + publish (gate wave)
            pc[4] += 1;
            pc[5] += prof_spins; // failed polls
        }
    }
    if (gate_wave && (l & 3) == 0) // lstm.cpp:160-161: the state carries into the next segment
    {
        a.state[state_off(target, a.layer, dir, 0, Hl) + unit] = hlast;
        a.state[state_off(target, a.layer, dir, 1, Hl) + unit] = c;
    }
    if (prof && l == 0)
        for (int i = 0; i < 6; ++i)
            a.prof[(a.layer * 2 + (w == gw ? 0 : 1)) * 8 + i] = pc[i];
}

// grid = 8*S workgroups (1-D), cooperative launch.  Roles come from the census (see file header).
// Must stay <= 120 VGPRs: see gemm_kernels.h (register budget of the two-slot pipeline)
template <int KPW, bool PRECISE> __global__ __launch_bounds__(LSTM_PERSISTENT_THREADS) void lstm_persistent_kernel(LstmArgs a)
{
    __shared__ float part[2][8][64];
    __shared__ float pbuf[LSTM_P_RING][64]; // ring of W_ih x + b_ih rows (see lstm_persistent_body)
    __shared__ int s_ctl[4]; // chain, slice, fast, abort
    const int tid = threadIdx.x;
    const int nwg = gridDim.x, S = a.S;
    if (tid == 0)
        lstm_census(a.sync, a.status, S, nwg, a.force_safe, s_ctl);
    __syncthreads();
    const int chain = s_ctl[0], slice = s_ctl[1];
    if (a.prof && tid == 0) // placement record of the profiler: where this workgroup runs and which role it took
    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        a.prof[64 + blockIdx.x] = ((unsigned long long)(xcc_id() & 15) << 48) | ((unsigned long long)(chain & 0xff) << 40) |
                                  ((unsigned long long)(slice & 0xff) << 32) | hw;
    }
    if (s_ctl[3] || chain >= a.nchains) // aborted, or an XCD / block range with no chain to run
        return;
    if (s_ctl[2])
        lstm_persistent_body<KPW, true, PRECISE>(a, chain, slice, part, pbuf, &s_ctl[3]);
    else
        lstm_persistent_body<KPW, false, PRECISE>(a, chain, slice, part, pbuf, &s_ctl[3]);
}

} // namespace umx
