// lstm_kernels.h -- the 3-layer bidirectional streaming LSTM recurrence (lstm.cpp:101-179).
//
// The input projection W_ih x_t + b_ih for all frames is a batched MFMA GEMM (gemm_kernels.h,
// mode G_IH) -- legal because it has no recurrence.  What is left per (target, layer, dir)
// "chain" is strictly serial:  gates_t = (P_t + W_hh h_{t-1}) + b_hh, i|f|g|o split
// (lstm.cpp:143-152), c = sig(f) c + sig(i) tanh(g), h = sig(o) tanh(c).
// 4 targets x 2 directions = 8 independent chains per layer run concurrently.
//
// Work split: one workgroup (512 threads, 8 waves) per (chain, slice of 16 hidden units).
// A slice owns the 64 gate columns {g*Hl + 16*slice + u}; lane l of every wave is gate column
// l = g*16 + u; wave w owns the k-range [w*Hl/8, (w+1)*Hl/8) of the W_hh . h contraction.
//   W    float [chains][S][Hl][64]     (k-major: a wave reads 64 consecutive floats per k)
//   bhh  float [chains][S][64]
//   P    float [Tp][2][S][64] per target (GEMM output, columns permuted to match)
// For Hl = 512: S = 32 slices -> 256 workgroups, W slice = 128 KiB fp32 = 64 VGPRs per lane.
//
// Two drivers share the same per-step arithmetic (bitwise-identical results):
//   lstm_step_kernel        one launch per timestep, h/c in HBM between launches (simple, safe)
//   lstm_persistent_kernel  one launch per layer: W_hh stays in VGPRs for all T steps, h is
//                           exchanged between the chain's workgroups through 8-byte
//                           {tag, value} granules written with agent-scope (sc1, write-through)
//                           stores and polled with agent-scope relaxed loads -- the
//                           placement-independent "data is the flag" hand-off.  Each wave
//                           polls exactly the Hl/8 (<= 64) granules of its own k-range, one per
//                           lane, and broadcasts them with v_readlane, so h never goes
//                           through LDS.  Two granule slots (step parity) are enough: a
//                           producer can only overwrite slot p two steps later, which needs
//                           every consumer's next h, i.e. every consumer is past its reads.
// Floor for one segment: 3 layers x 2584 steps = 7752 serially dependent steps.
#pragma once
#include "common.h"

namespace umx
{

struct LstmArgs
{
    const float *W;   // this layer: [chains][S][Hl][64]
    const float *bhh; // [chains][S][64]
    const float *P[4];
    float *out[4];    // out[target][t*ldo + col0 + dir*Hl + unit]
    float *state;     // [4 targets][3 layers][2 dirs][2 (h,c)][Hl]   (lstm.hpp:10-16 h, c)
    float *hbuf;      // step driver: [2][chains][Hl] ping-pong h
    unsigned long long *granules; // persistent driver: [2][chains][Hl]
    unsigned *status; // [0] = abort/timeout flag (0 = ok)
    int Hl, S, T, ldp, ldo, col0, layer;
    int tmap[4];      // grid chain>>1 -> target (targets can be skipped: BASELINE config 1)
};

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); } // lstm.cpp:36-39

// pre-activation (all 64 lanes of wave 0) -> new (c, h) on lanes 0..15
__device__ __forceinline__ void lstm_cell(float pre, int lane, float &c, float &h)
{
    const int g = lane >> 4, u = lane & 15;
    const float act = (g == 2) ? tanhf(pre) : sigmoid_ref(pre);
    const float i_t = __shfl(act, u), f_t = __shfl(act, 16 + u), g_t = __shfl(act, 32 + u),
                o_t = __shfl(act, 48 + u);
    const float c_t = f_t * c + i_t * g_t; // lstm.cpp:154-156
    c = c_t;
    h = o_t * tanhf(c_t); // lstm.cpp:157
}

__device__ __forceinline__ size_t state_off(int target, int layer, int dir, int hc, int Hl)
{
    return ((((size_t)target * 3 + layer) * 2 + dir) * 2 + hc) * Hl;
}

// ------------------------------------------------------------------ per-step driver
// grid (S, chains); hbuf[step&1] holds h_{t-1}, hbuf[(step+1)&1] receives h_t; c lives in state.
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_kernel(LstmArgs a, int step)
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
    const float *Wp = a.W + (((size_t)wchain * a.S + slice) * Hl + (size_t)w * kpw) * 64 + l;
    float acc = 0.f;
    for (int i = 0; i < kpw; ++i)
        acc = fmaf(Wp[(size_t)i * 64], hs[w * kpw + i], acc);
    part[w][l] = acc;
    __syncthreads();
    if (w == 0)
    {
        float s = part[0][l];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww)
            s += part[ww][l];
        const float p = a.P[target][(size_t)t * a.ldp + ((size_t)dir * a.S + slice) * 64 + l];
        const float pre = (p + s) + a.bhh[((size_t)wchain * a.S + slice) * 64 + l]; // lstm.cpp:132-140
        const int unit = slice * 16 + (l & 15);
        float *cst = a.state + state_off(target, a.layer, dir, 1, Hl);
        float c = (l < 16) ? cst[unit] : 0.f, h;
        lstm_cell(pre, l, c, h);
        if (l < 16)
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

constexpr unsigned LSTM_SPIN_LIMIT = 1u << 22; // bounded spin: ~seconds, then abort the launch

// KPW = Hl/8 = k-range (and granules) per wave; KPW <= 64.
template <int KPW> __global__ __launch_bounds__(LSTM_THREADS) void lstm_persistent_kernel(LstmArgs a)
{
    __shared__ float part[2][8][64];
    __shared__ int abort_flag;
    const int slice = blockIdx.x, chain = blockIdx.y, target = a.tmap[chain >> 1], dir = chain & 1;
    const int wchain = target * 2 + dir;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    constexpr int Hl = KPW * 8;
    const int nchains = gridDim.y, T = a.T;

    // W_hh slice -> registers, once
    float W[KPW];
    {
        const float *Wp = a.W + (((size_t)wchain * a.S + slice) * Hl + (size_t)w * KPW) * 64 + l;
#pragma unroll
        for (int i = 0; i < KPW; ++i)
            W[i] = Wp[(size_t)i * 64];
    }
    const float bh = a.bhh[((size_t)wchain * a.S + slice) * 64 + l];
    const int unit = slice * 16 + (l & 15);
    float c = 0.f;
    if (w == 0 && l < 16)
        c = a.state[state_off(target, a.layer, dir, 1, Hl) + unit];
    // h_{-1}: this wave's k-range, one value per lane (lanes >= KPW idle)
    float hval = 0.f;
    if (l < KPW)
        hval = a.state[state_off(target, a.layer, dir, 0, Hl) + w * KPW + l];
    if (tid == 0)
        abort_flag = 0;
    __syncthreads();

    gu64 *gran = (gu64 *)a.granules;
    gu32 *status = (gu32 *)a.status;
    const float *Pp = a.P[target] + ((size_t)dir * a.S + slice) * 64 + l;
    float hlast = 0.f;

    for (int step = 0; step < T; ++step)
    {
        const int t = dir == 0 ? step : T - 1 - step;
        float p = 0.f;
        if (w == 0)
            p = Pp[(size_t)t * a.ldp]; // issued before the poll: latency hides behind it
        if (step > 0)
        {
            // wait for h_{step-1}: tag == step, slot (step-1)&1
            gu64 *g = gran + ((size_t)((step - 1) & 1) * nchains + chain) * Hl + w * KPW + l;
            const unsigned want = (unsigned)step;
            bool ok = (l >= KPW);
            unsigned long long x = 0;
            unsigned spins = 0;
            for (;;)
            {
                if (!ok)
                {
                    x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)(x >> 32) == want;
                }
                if (__all(ok))
                    break;
                if (++spins > LSTM_SPIN_LIMIT ||
                    ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                {
                    if (l == 0)
                        __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    abort_flag = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            hval = __uint_as_float((unsigned)x);
        }
        // partial dot product over this wave's k-range, h broadcast lane -> SGPR
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KPW; ++i)
        {
            const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hval), i));
            acc = fmaf(W[i], hk, acc);
        }
        part[step & 1][w][l] = acc;
        __syncthreads();
        if (abort_flag) // uniform after the barrier: every wave leaves, nobody is left spinning on us
            return;
        if (w == 0)
        {
            float s = part[step & 1][0][l];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww)
                s += part[step & 1][ww][l];
            const float pre = (p + s) + bh;
            float h;
            lstm_cell(pre, l, c, h);
            if (l < 16)
            {
                const unsigned long long gv =
                    ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned long long)__float_as_uint(h);
                __hip_atomic_store(gran + ((size_t)(step & 1) * nchains + chain) * Hl + unit, gv,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.out[target][(size_t)t * a.ldo + a.col0 + dir * Hl + unit] = h;
                hlast = h;
            }
        }
    }
    if (w == 0 && l < 16) // lstm.cpp:160-161: the state carries into the next segment
    {
        a.state[state_off(target, a.layer, dir, 0, Hl) + unit] = hlast;
        a.state[state_off(target, a.layer, dir, 1, Hl) + unit] = c;
    }
}

} // namespace umx
