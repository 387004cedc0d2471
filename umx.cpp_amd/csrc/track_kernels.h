// track_kernels.h -- on-device overlap-add of split_inference (umx.cpp:197-273): the whole track stays in HBM,
// each segment's stems are blended into the track accumulators with the triangular transition weight and the
// sum of weights normalises at the end.  Same fp32 operations in the same order as host/split.cpp (which
// follows the reference), so the two drivers agree bit for bit.
#pragma once
#include "common.h"

namespace umx
{

struct Stems4
{
    float2 *p[4]; // (2,n) interleaved = one float2 per sample
};

// umx.cpp:197-206, 246: weight[i] = weight[N-1-i] = i+1 (i < N/2), divided by the maximum, ^1.0
__device__ __forceinline__ float transition_weight(int k, int N)
{
    const float raw = (float)((k < N / 2) ? k + 1 : N - k);
    return raw / (float)(N / 2);
}

// umx.cpp:234-260: out(:, offset+k) += w(k) * chunk_out(:, k); sum_weight(offset+k) += w(k)
__global__ void track_accumulate_kernel(Stems4 track, float *sum_w, Stems4 seg, int offset, int n, int N)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (k >= n)
        return;
    const float w = transition_weight(k, N);
    const float2 s = seg.p[t][k];
    float2 o = track.p[t][(size_t)offset + k];
    o.x += w * s.x;
    o.y += w * s.y;
    track.p[t][(size_t)offset + k] = o;
    if (t == 0)
        sum_w[(size_t)offset + k] += w;
}

// the two halves of that update for a multi-GPU driver: the rank that ran the segment multiplies (w * chunk, one
// rounding), the rank that owns the track adds (a second rounding) -- the same two fp32 operations as above
__global__ void track_weight_kernel(Stems4 seg, int n, int N)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (k >= n)
        return;
    const float w = transition_weight(k, N);
    float2 s = seg.p[t][k];
    s.x = w * s.x;
    s.y = w * s.y;
    seg.p[t][k] = s;
}
__global__ void track_add_weighted_kernel(Stems4 track, float *sum_w, Stems4 wseg, int offset, int n, int N)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (k >= n)
        return;
    const float2 s = wseg.p[t][k];
    float2 o = track.p[t][(size_t)offset + k];
    o.x += s.x;
    o.y += s.y;
    track.p[t][(size_t)offset + k] = o;
    if (t == 0)
        sum_w[(size_t)offset + k] += transition_weight(k, N);
}

// umx.cpp:264-273: out /= sum_weight, for samples [start, start + count)
__global__ void track_normalise_kernel(Stems4 track, const float *sum_w, int start, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (i >= count)
        return;
    const size_t k = (size_t)start + i;
    const float sw = sum_w[k];
    float2 o = track.p[t][k];
    o.x /= sw;
    o.y /= sw;
    track.p[t][k] = o;
}

} // namespace umx
