// lstm_batch.h -- the streaming LSTM recurrence (lstm.cpp:101-179) for a BATCH of up to 16 independent tracks
// (SURVEY 8f-4).  Same chains, same sharding and the same granule hand-off as lstm_kernels.h, but the per-step
// matrix-vector product W_hh h of one track becomes the matrix-matrix product
//         gates[64 gate columns of this workgroup][16 tracks] = W_hh-slice [64 x Hl] . H [Hl x 16 tracks]
// which is what the matrix cores are for: v_mfma_f32_16x16x32_bf16 with fp32 accumulation and the three-term bf16
// split of gemm_bf16x3.h, so the arithmetic stays fp32-class.  One serial step -- hand-off, barrier,
// gates -- now advances every track of the batch, which is the only way to lift a latency-bound recurrence.
//
// Work split (unchanged): workgroup (chain, slice) owns 16 hidden units = 64 gate columns (column = 4*unit + gate,
// gates i|f|g|o, lstm.cpp:143-152); 512 threads = 8 waves; wave w owns the k-range [w*Hl/8, (w+1)*Hl/8) of the
// contraction.
//   A operand = W_hh: M = gate column (4 tiles of 16), kept in VGPRs for the whole launch
//       general form (fp32 or dequantised weights): three bf16 planes, six products a1b1+a1b2+a2b1+a2b2+a1b3+a3b1
//       u8-resident W_hh (the ggml file's own storage, the default): q - 128 is an integer in [-128, 127] and therefore
//       EXACT in bf16: one plane, three products (q-128).(h1+h2+h3), and the affine map of model.cpp:610-616 is
//       applied to the sum:  W_hh h = scale * sum_k (q_k - 128) h_k + (offset + 128 scale) * sum_k h_k.  The second
//       sum comes out of a fifth M tile whose A operand is all ones.  (The reference rounds q*scale+offset to fp32
//       per weight first; the two differ by that rounding, ~1e-7 of the dot product, the size of one fp32 rounding
//       of the sum itself.)
//   B operand = h_{t-1}: N = track (lane & 15), K = hidden unit.  The producer publishes h ALREADY SPLIT and already
//       in fragment order: one 16-byte granule per (PAIR of units 2i, 2i+1; track) = {tag:32, h1 pair, h2 pair, h3 pair}
//       (h = h1 + h2 + h3 exactly, bf16 terms; the two units of a pair sit in lanes 16 apart of the gate wave and meet
//       through one ds_swizzle) -- 8 bytes per value like lstm_kernels.h's {tag, fp32}, still ONE naturally aligned
//       store that is its own flag (16-byte stores are observed untorn on gfx950, MI355X_MICROARCH.md), and the three
//       payload dwords of the consumer's i-th load ARE dword i of its three B fragments: no unpacking at all.  A lane fetches its B fragment (8 units of
//       one track) with four 16-byte loads per K step, each of them contiguous across the wave (lstmb_granule_index).
//   C = the 4 gates of unit 4*tile + (lane>>4) for track lane&15 sit in ONE lane (rows of the 16x16 result are
//       4*(lane>>4)+reg): no cross-lane traffic in the gate phase.  The eight k-range partials meet through LDS
//       (float4 per lane, conflict-free), summed in a fixed tree: results never depend on the batch size or on
//       which lanes of the batch are active (an absent track is a zero column of B).
// Everything else -- census / intra-XCD vs sc1 protocol, bounded spins, bulk W_ih-row ring through global_load_lds,
// the in-kernel profiler -- follows lstm_kernels.h.  A launch covers the steps [t_begin, t_end) and starts from /
// leaves behind the fp32 (h, c) stream state, so the same kernel run one step per launch is the bit-identical
// fallback (and cross-check) that needs no co-residency.
#pragma once
#include "gemm_bf16x3.h"
#include "gemm_planes.h" // split2_f16
#include "lstm_kernels.h"

namespace umx
{

static_assert(MAX_TRACK_LANES == 64, "common.h LaneSet covers every track lane of a context");
constexpr int LSTMB_MAX_TRACKS = 64;  // track lanes per context: 16 per launch of lstm_batch_kernel (more: one launch per group of 16), 32 / 64 per launch of lstm_batch8_kernel
constexpr int LSTMB_GROUP_TRACKS = 16; // the matrix instruction's N
// x64 shader cycles a wave sleeps before its first poll of a step.  A wave that also runs the gate phase has just
// published and needs one hand-off latency; the other waves come straight from the barrier and have the whole gate
// phase in front of them -- polling through it would only load the L2 (every failed attempt is 8 x 16 B per lane)
// and take issue slots from the gate wave sharing their SIMD.
#define LSTMB_DELAY_GATE 0
#define LSTMB_DELAY_IDLE 24
#define LSTMB_PROF_WAVE 4 // the second wave the in-kernel profiler reports (beside wave 0): 4..7 = a wave without gate work
#define LSTMB_RETRY_SLEEP 1 // x64 cycles between failed polls
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct LstmBArgs
{
    const float *W;          // this layer: [chains][S][Hl][64] fp32 ...
    const unsigned char *Wq; // ... or (W == nullptr) the ggml file's u8 in the same layout
    float wsc[8], wof[8];    // per weight chain: scale, offset (model.cpp:610-616)
    const float *bhh;        // [chains][S][64]
    const float *P[4];       // track lane n of target i: P[i] + n * p_stride
    float *out[4];           // out[i] + n * out_stride + t*ldo + col0 + dir*Hl + unit
    float *state;            // lane n: state + n * state_stride, then [4 targets][3 layers][2 dirs][2 (h,c)][Hl]
    float *state_out;        // where the launch leaves h / c: `state` itself for a launch that covers the whole segment; the OTHER
                             // of two copies for the one-step launches of the per-step driver -- a workgroup reads the h of its whole
                             // chain when it starts and writes its own 16 units when it ends, and nothing keeps a late workgroup
                             // of a launch from starting after an early one has finished
    size_t p_stride, out_stride, state_stride;
    unsigned *sync;          // [0..7] census, [8] arrivals, [LSTM_SYNC_HEADER_WORDS..] granules u64 [2][8][Hl/8][16][8]
    unsigned *status;
    unsigned long long *prof;
    int Hl, S, T, ldp, ldo, col0, layer, nchains;
    int tmap[4];
    int force_safe;
    unsigned tag_epoch; // a granule's tag is (epoch << 12) | (step + 1): unique per launch (20 bits of epoch)
    unsigned long long lane_mask; // bit n = track lane n takes part in this launch
    int nbp;            // lanes the LDS arrays are sized for: power of two >= highest active lane + 1
    int bulk;           // W_ih-row ring: rows per bulk fetch (ring = 2 * bulk rows)
    int t_begin, t_end; // steps of this launch
    int abort_at;       // testing: every workgroup gives up at this step as if a poll had timed out (0 = never)
    int census;         // 1: take the census (t_end - t_begin > 1: needs the grid co-resident); 0: static roles
    // Round 4: the recurrence writes the NEXT GEMM's A operand itself (u8-resident W_hh only).  The gate lanes hold h as the two
    // fp16 planes of h * 2^14 anyway (the granule's payload = exactly what split_planes_kernel would produce from the fp32 row), and
    // sum_k h'_k over a chain's 512 units -- the row sum the consumer's affine fix-up needs -- is what the all-ones matrix tile of
    // every workgroup computes one step later.  planes[i] = nullptr: not fused (the fp32 rows are split by split_planes_kernel).
    unsigned short *planes[4]; // target i: [2][rows][ldpl] fp16 bits; row = lane * Tp + frame, column = col0 + dir * Hl + unit
    size_t plane_elems;        // elements between the two planes
    float *rs_dir[4];          // target i: [2 dirs][rs_rows] row sums of this layer's output per direction (sum_k a_k, unscaled)
    size_t rs_rows;
    int ldpl, Tp;
    int write_f32;             // with planes: also keep the fp32 rows of every step (debug taps); without planes they always are
    int lane_base;             // lstm_batch8.h: the launch serves track lanes [lane_base, lane_base + 32)
    int poll_delay;            // lstm_batch8.h: x64 cycles a wave sleeps between its publication and its first poll of the next step
};

__host__ __device__ inline size_t lstmb_granule_words(int Hl) { return (size_t)2 * 8 * Hl * 16 * 2; } // 32-bit words
// index (in 16-byte granules) of the granule holding hidden units (k & ~1, k | 1) of (step parity slot, chain, track n).
// Within a K step (32 units) the order is [pair i = (k%8)/2][8-unit group q = (k%32)/8][track]: the consumer's i-th
// 16-byte load is then CONTIGUOUS across the wave (lane = q*16 + n), one L2 request per 128 B, and the tracks of a
// launch are packed (nbp), so small batches do not spread over sixteen times the lines.
__host__ __device__ inline size_t lstmb_granule_index(int slot, int chain, int k, int n, int Hl, int nbp)
{
    return ((((size_t)(slot * 8 + chain) * (Hl / 32) + (k >> 5)) * 4 + ((k & 7) >> 1)) * 4 + ((k & 31) >> 3)) * nbp + n;
}
// 16-byte granule store through the granule area's buffer resource.  FAST (all parties share one XCD L2): plain
// store; SAFE: write-through (sc0 sc1).
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
template <bool FAST> __device__ __forceinline__ void granule_store16(__amdgpu_buffer_rsrc_t rs, int byte_offset, uint4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), rs, byte_offset, 0, FAST ? 0 : 17);
}
// One (row, track) of the W_ih x + b_ih ring: 64 gate columns = 256 bytes, stored every LSTMB_RING_PITCH bytes.  The gate
// lanes of a 16-lane read group differ in the TRACK and read the same 16 bytes of 16 different (row, track) blocks: at a
// pitch of 256 bytes all of them hit the same four banks (a 16-way conflict: 21 % of the LDS cycles of round 2's kernel,
// the only kernel of the profile with any); 16 bytes of padding spread them over all 64 banks.
#define LSTMB_RING_PITCH_BYTES 272
constexpr int LSTMB_RING_PITCH = LSTMB_RING_PITCH_BYTES;
constexpr size_t LSTMB_HSW_BYTES = 2 * 8 * 16 * sizeof(float); // sum_k h'_k per k-range and lane, two steps (fused row sums): the last KiB
__host__ __device__ inline size_t lstmb_lds_bytes(int nbp, int bulk, int sp = 1) // sp: slice span of the workgroup (lstmb_body)
{
    return (size_t)2 * 8 * 16 * sp * nbp * 16 /* part */ + (sp > 1 ? (size_t)0 : (size_t)2 * bulk * nbp * LSTMB_RING_PITCH) /* ring (sp = 1) */ +
           LSTMB_HSW_BYTES;
}

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 &v) { return __builtin_bit_cast(bf16x8, v); }

// sum of the NDW k-range partials in a fixed tree
template <int NDW> __device__ __forceinline__ float tree_sum(const float (&p)[8])
{
    if (NDW == 8)
        return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    if (NDW == 4)
        return (p[0] + p[1]) + (p[2] + p[3]);
    if (NDW == 2)
        return p[0] + p[1];
    return p[0];
}

template <int NDW> __device__ __forceinline__ float2v tree_sum2(const float2v (&p)[8])
{
    if (NDW == 8)
        return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    if (NDW == 4)
        return (p[0] + p[1]) + (p[2] + p[3]);
    if (NDW == 2)
        return p[0] + p[1];
    return p[0];
}

// WQ form: sum_k h'_k over a wave's k-range (the factor of the weight tensor's offset, model.cpp:610-616 applied to the sum) comes
// out of a fifth M tile whose A operand is all ones -- 4 of a wave's 20 matrix instructions.  (Round 3 measured the alternative:
// the pair sums (h1 + h2) + (h1' + h2') travelling in the granules' free fourth dword, added by the consumer and folded over the
// four 8-unit groups with two lane exchanges: 20 % fewer matrix-pipe cycles, 1.5 % SLOWER -- two dependent lane exchanges on the
// turn's critical path cost more than four queued matrix instructions on an idle pipe.  Removed in round 4; the dword is zero.)
// SP = slice span: the workgroup owns SP x 16 hidden units = SP x 64 gate columns (SP = 1: the form described above, the only one
// instantiated since round 6; SP = 2 was round 4's side-by-side kernel -- eight M tiles, every wave also a gate wave).  group: the workgroup serves lanes
// [16 group, 16 group + 16) of the launch, with a granule area of their own.
template <int HL, bool WQ, bool FAST, bool PRECISE, int SP = 1>
__device__ __forceinline__ void lstmb_body(const LstmBArgs &a, int chain, int slice, unsigned char *smem, int *abort_flag, int group = 0)
{
    const int lane0 = LSTMB_GROUP_TRACKS * group;
    constexpr int NKS = HL / 32;                 // K steps (32 hidden units each) of the whole contraction
    constexpr int KSW = NKS >= 8 ? NKS / 8 : 1;  // K steps per dot wave
    constexpr int NDW = NKS >= 8 ? 8 : NKS;      // waves that multiply (all 8 for Hl >= 256)
    constexpr int MT = 4 * SP;                   // M tiles (16 gate columns = 4 hidden units each) of the workgroup
    static_assert(SP == 1 || (SP == 2 && NKS >= 8 && WQ), "slice span 2: u8-resident W_hh, eight multiply waves");
    constexpr int NPL = WQ ? 1 : 3;              // bf16 planes of W_hh held in registers
    const int target = a.tmap[chain >> 1], dir = chain & 1, wchain = target * 2 + dir;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 15, q = l >> 4;
    const int nbp = SP > 1 ? LSTMB_GROUP_TRACKS : a.nbp, bulk = a.bulk, ring_mask = 2 * bulk - 1, T = a.T, S = a.S;
    const unsigned lane_mask = (unsigned)(a.lane_mask >> lane0) & 0xffffu;
    const bool lane_on = (lane_mask >> n) & 1u;
    const bool dot_wave = w < NDW, gate_wave = w < MT; // gate wave w finishes M tile w (units 4w .. 4w+3 of the workgroup's)
    constexpr int RING_PITCH = LSTMB_RING_PITCH;

    float4 *part = reinterpret_cast<float4 *>(smem);                                   // [2][8 waves][MT tiles][4 q][nbp]
    unsigned char *ring = smem + (size_t)2 * 8 * 4 * MT * nbp * 16;                   // [2*bulk rows][nbp] blocks of 64 SP floats, RING_PITCH apart

    // ---- W_hh fragments: lane (i = l & 15, q) of tile mt holds gate column 16 mt + i, units k = 32 ks' + 8 q + j
    bf16x8 Wf[MT][KSW][NPL];
    if (dot_wave)
    {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
            {
                float wv[8];
                // the weights stay in slices of 64 gate columns ([chain][S][Hl][64]): tile mt is tile mt & 3 of slice SP slice + mt / 4
                const size_t base = (((size_t)wchain * S + slice * SP + (mt >> 2)) * HL + (size_t)(w * KSW + ks) * 32 + 8 * q) * 64 + 16 * (mt & 3) + n;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    wv[j] = WQ ? (float)a.Wq[base + (size_t)j * 64] - 128.0f : whh_at(a.W, a.Wq, a.wsc[wchain], a.wof[wchain], base + (size_t)j * 64);
                uint4 p1, p2, p3;
                if (WQ) // an integer in [-128, 127]: exact in ONE fp16 plane
                {
                    f16x8 hw;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        hw[j] = (_Float16)wv[j];
                    p1 = __builtin_bit_cast(uint4, hw);
                }
                else
                    split3(wv, p1, p2, p3);
                Wf[mt][ks][0] = as_bf16x8(p1);
                if (!WQ)
                {
                    Wf[mt][ks][NPL > 1 ? 1 : 0] = as_bf16x8(p2);
                    Wf[mt][ks][NPL > 2 ? 2 : 0] = as_bf16x8(p3);
                }
            }
    }
    // WQ: h travels as TWO fp16 planes of h * 2^14 (h1 = fp16(h'), h2 = fp16(h' - h1): 22 significand bits + the
    // residual's sign; |h| < 1 so h' < 2^14, and the second plane is a normal fp16 number down to residuals of 2^-28)
    // against ONE exact fp16 plane of q - 128: two products instead of the three of a bf16 split.  The power of two
    // comes back out with the scale:  W h = (wsc 2^-14) * sum (q-128) h' + ((wof + 128 wsc) 2^-14) * sum h'
    constexpr float HSCALE = 16384.0f;
    const float wsc = a.wsc[wchain] * (WQ ? 1.0f / HSCALE : 1.0f),
                wof2 = (a.wof[wchain] + 128.0f * a.wsc[wchain]) * (WQ ? 1.0f / HSCALE : 1.0f);

    // ---- per-(unit, track) cell state of the gate lanes, b_hh of the unit's four gates
    const int unit = slice * 16 * SP + 4 * (w & (MT - 1)) + q;
    const size_t st_h = (size_t)(lane0 + n) * a.state_stride + state_off(target, a.layer, dir, 0, HL);
    const size_t st_c = (size_t)(lane0 + n) * a.state_stride + state_off(target, a.layer, dir, 1, HL);
    float c = 0.f, hlast = 0.f;
    float4 bh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gate_wave)
    {
        bh = *reinterpret_cast<const float4 *>(a.bhh + ((size_t)wchain * S + slice * SP + (w >> 2)) * 64 + 4 * (4 * (w & 3) + q));
        if (lane_on)
        {
            c = a.state[st_c + unit];
            hlast = a.state[st_h + unit];
        }
    }
    // ---- h_{t_begin - 1} from the fp32 stream state, split like a published granule
    bf16x8 hf[KSW][3];
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks)
    {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            hv[j] = (dot_wave && lane_on) ? a.state[st_h + (w * KSW + ks) * 32 + 8 * q + j] : 0.f;
        uint4 p1, p2, p3 = make_uint4(0u, 0u, 0u, 0u);
        if (WQ)
            split2_f16(hv, HSCALE, p1, p2);
        else
            split3(hv, p1, p2, p3);
        hf[ks][0] = as_bf16x8(p1);
        hf[ks][1] = as_bf16x8(p2);
        hf[ks][2] = as_bf16x8(p3);
    }

    // the polls are 16-byte L1-bypassing buffer loads (buffer_load_dwordx4 ... sc1): two granules each
    const __amdgpu_buffer_rsrc_t gran_rs =
        __builtin_amdgcn_make_buffer_rsrc(a.sync + LSTM_SYNC_HEADER_WORDS + (size_t)group * lstmb_granule_words(HL), 0, (int)(lstmb_granule_words(HL) * 4), 0x00020000);
    gu32 *status = (gu32 *)a.status;
    const float *const Pp = a.P[target] + (size_t)lane0 * a.p_stride + ((size_t)dir * S + slice * SP) * 64 + l;
    float *const outp = a.out[target] + (size_t)(lane0 + n) * a.out_stride + a.col0 + dir * HL + unit;
    const size_t ldp = (size_t)a.ldp, ldo = (size_t)a.ldo, p_stride = a.p_stride;
    const unsigned tag_hi = a.tag_epoch << 12;
    const int t_begin = a.t_begin, t_end = a.t_end;
    // fused A planes of the consumer (LstmBArgs::planes): this lane's column of its track lane's rows; row sums by workgroup 0 of a chain
    unsigned short *const plp = (WQ && a.planes[target]) ? a.planes[target] + (size_t)(lane0 + n) * a.Tp * a.ldpl + a.col0 + dir * HL + unit : nullptr;
    const size_t plane_elems = a.plane_elems, ldpl = (size_t)a.ldpl;
    float *const rsp = (WQ && a.rs_dir[target] && slice == 0) ? a.rs_dir[target] + (size_t)dir * a.rs_rows + (size_t)lane0 * a.Tp : nullptr;
    float *const hsw = reinterpret_cast<float *>(smem + lstmb_lds_bytes(nbp, bulk, SP) - LSTMB_HSW_BYTES); // [2][8 waves][16]: sum_k h'_k per k-range and lane
    unsigned plast = 0; // the fp16 planes of hlast (h1 | h2 << 16)
    // the row sum of the row that step `sm` multiplied with (h'_{sm-1}: frame sm - 1 forward, T - sm backward; at step 0 that is the
    // carried state, not a row): the eight k-ranges' sums in a fixed tree, by sixteen lanes of the last multiply wave of the chain's
    // workgroup 0 -- WHILE that wave waits for the next step's polls (or behind the loop), not on the hand-off's path: as a tail of
    // the gate phase it made workgroup 0 the slowest producer of its chain and cost 6 % (19 % with the groups in turn).  The row of
    // the launch's last step has no later step: lstm_last_row_sum_kernel.
    auto row_sum_of_step = [&](int sm) {
        if (rsp && w == NDW - 1 && l < nbp && sm > 0 && ((lane_mask >> l) & 1u))
        {
            float hp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < NDW; ++ww)
                hp[ww] = hsw[((sm & 1) * 8 + ww) * 16 + l];
            rsp[(size_t)l * a.Tp + (size_t)(dir == 0 ? sm - 1 : T - sm)] = tree_sum<NDW>(hp) * (1.0f / 16384.0f);
        }
    };

    // W_ih x + b_ih rows: bulk fetch into the LDS ring (see lstm_kernels.h for why), rows x active lanes dealt to the
    // dot waves; a wave-instruction moves the 64 columns of one (row, track)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    typedef const __attribute__((address_space(1))) void *glb_ptr;
    const unsigned ring_lds = (unsigned)(size_t)(lds_ptr)ring;
    auto fetch_rows = [&](int first_row) {
        const int items = bulk * nbp;
        for (int i = w; i < items; i += NDW)
        {
            const int r = first_row + i / nbp, nn = i % nbp;
            if (r < t_end && ((lane_mask >> nn) & 1u))
                __builtin_amdgcn_global_load_lds((glb_ptr)(Pp + (size_t)nn * p_stride + (size_t)(dir == 0 ? r : T - 1 - r) * ldp),
                                                 (lds_ptr)(size_t)(ring_lds + (unsigned)RING_PITCH * (unsigned)((r & ring_mask) * nbp + nn)), 4, 0, 0);
        }
    };
    // SP > 1 (every wave multiplies AND finishes a tile): no ring -- a gate lane fetches the 16 bytes of its own (unit, track) of row
    // step + 1 straight into registers right behind the polls of step `step`: a whole step ahead of its use, in front of the next
    // step's polls in the memory queue (which return in order: the row is there when they are), no LDS and no other wave involved
    const float *const Pg = a.P[target] + (size_t)(lane0 + n) * a.p_stride + ((size_t)dir * S + slice * SP + (w >> 2)) * 64 + 4 * (4 * (w & 3) + q);
    float4 p4n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SP > 1)
    {
        if (gate_wave && lane_on && t_begin < t_end)
            p4n = *reinterpret_cast<const float4 *>(Pg + (size_t)(dir == 0 ? t_begin : T - 1 - t_begin) * ldp);
    }
    else if (dot_wave)
    {
        fetch_rows(t_begin);
        fetch_rows(t_begin + bulk);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
    }
    // vmcnt(0) for EVERY wave: what was loaded above (bias, state, weights) is then known to have arrived, and the compiler does
    // not place a "wait for everything" in front of the bias add of the gate phase -- which, inside the loop, is a wait for the
    // acknowledgements of the row / plane stores issued a few hundred cycles earlier, on the hand-off's critical path
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads(); // the first rows are read before the first step's barrier
    const bool prof = a.prof != nullptr && group == 0 && chain == 0 && slice == 0 && (w == 0 || w == LSTMB_PROF_WAVE);
    const int pw_idx = w == 0 ? 0 : 1;
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, pc6 = 0, pc7 = 0;
    unsigned prof_spins = 0;

    for (int step = t_begin; step < t_end; ++step)
    {
        long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        float4 p4s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prof)
            c0 = clock64();
        if (a.abort_at && step == a.abort_at && tid == 0)
        {
            __hip_atomic_store(status, 1u + (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *abort_flag = 1;
        }
        if (dot_wave)
        {
            if (step > t_begin)
            {
                // h_{step-1}: granules of slot (step-1)&1 tagged `step`; this lane's 8 units x KSW K steps of track n
                const unsigned want = tag_hi | (unsigned)step;
                int goff[KSW]; // byte offset of this lane's first 16 B (units 0,1 of its 8) inside the granule area
#pragma unroll
                for (int ks = 0; ks < KSW; ++ks)
                    goff[ks] = (int)(lstmb_granule_index((step - 1) & 1, chain, (w * KSW + ks) * 32 + 8 * q, n, HL, nbp) * 16);
                if (FAST)
                {
                    if (gate_wave)
                        __builtin_amdgcn_s_sleep(LSTMB_DELAY_GATE);
                    else
                        __builtin_amdgcn_s_sleep(LSTMB_DELAY_IDLE);
                }
                uint4 v[KSW][4];
                unsigned spins = 0;
                long long cs = 0;
                if (prof)
                    cs = clock64();
                for (;;)
                {
                    bool ok = true;
                    if (lane_on)
                    {
#pragma unroll
                        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                v[ks][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(gran_rs, goff[ks] + i * 64 * nbp, 0, 16)); // sc1
                        if (spins == 0)
                            row_sum_of_step(step - 1); // (LDS reads and one store in the shadow of the loads just issued)
                        unsigned bad = 0;
#pragma unroll
                        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                bad |= v[ks][i].x ^ want;
                        ok = bad == 0;
                    }
                    if (prof && spins == 0)
                    {
                        const long long ce = clock64();
                        pc6 += (unsigned long long)(cs - c0); // sleep
                        pc7 += (unsigned long long)(ce - cs); // first round of loads
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
                // granule i = {tag, h1 of units (2i, 2i+1), h2 pair, h3 pair | WQ: 0}: the payload dwords are the fragments' dwords
#pragma unroll
                for (int ks = 0; ks < KSW; ++ks)
                {
                    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                    const uint4 g0 = lane_on ? v[ks][0] : z, g1 = lane_on ? v[ks][1] : z, g2 = lane_on ? v[ks][2] : z, g3 = lane_on ? v[ks][3] : z;
                    hf[ks][0] = as_bf16x8(make_uint4(g0.y, g1.y, g2.y, g3.y));
                    hf[ks][1] = as_bf16x8(make_uint4(g0.z, g1.z, g2.z, g3.z));
                    hf[ks][2] = as_bf16x8(make_uint4(g0.w, g1.w, g2.w, g3.w));
                }
                // every gate wave of this workgroup is past iteration step - 2 (it has crossed the barrier of step - 1),
                // so ring rows <= step - 2 may be replaced: rows [step-1+bulk, step-1+2 bulk) take the slots of
                // [step-1-bulk, step-1); they are first read bulk - 1 barriers from now
                if (SP == 1 && step - t_begin > bulk && ((step - t_begin) & (bulk - 1)) == (bulk > 1 ? 1 : 0))
                    fetch_rows(step - 1 + bulk);
            }
            if (prof)
                c1 = clock64();
            if (SP > 1)
            {
                p4s = p4n; // row `step`, requested a step ago
                if (gate_wave && lane_on && step + 1 < t_end)
                    p4n = *reinterpret_cast<const float4 *>(Pg + (size_t)(dir == 0 ? step + 1 : T - 2 - step) * ldp);
            }
            floatx4 acc[MT];
            floatx4 accH = {0.f, 0.f, 0.f, 0.f};
            const f16x8 ones16 = __builtin_bit_cast(f16x8, make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
#define LSTMB_TERM(PW, PH)                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < KSW; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)           \
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[mt][ks][PW], hf[ks][PH], acc[mt], 0, 0, 0);
#define LSTMB_TERM16(PH)                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < KSW; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)           \
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Wf[mt][ks][0]),                 \
                                                         __builtin_bit_cast(f16x8, hf[ks][PH]), acc[mt], 0, 0, 0);
            if (WQ)
            {
                // smaller term first
                LSTMB_TERM16(1)
                LSTMB_TERM16(0)
#pragma unroll
                for (int ph = 1; ph >= 0; --ph)
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks)
                        accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones16, __builtin_bit_cast(f16x8, hf[ks][ph]), accH, 0, 0, 0);
            }
            else
            {
                LSTMB_TERM(NPL > 2 ? 2 : 0, 0)
                LSTMB_TERM(0, 2)
                LSTMB_TERM(NPL > 1 ? 1 : 0, 1)
                LSTMB_TERM(NPL > 1 ? 1 : 0, 0)
                LSTMB_TERM(0, 1)
                LSTMB_TERM(0, 0)
            }
#undef LSTMB_TERM
#undef LSTMB_TERM16
            const float hsum_wave = WQ ? accH[0] : 0.f;
            if (rsp && q == 0) // sum over this wave's k-range of h'_{step-1} for lane n: every workgroup has it, workgroup 0 of the chain keeps it
                hsw[((step & 1) * 8 + w) * 16 + n] = hsum_wave;
            if (n < nbp)
            {
                float4 *pw = part + ((size_t)(((step & 1) * 8 + w) * MT) * 4 + q) * nbp + n;
                // WQ: this k-range's share of W_hh h = wsc * sum (q-128) h + (wof + 128 wsc) * sum h (every row of accH
                // holds the same sum of h)
                const float hs = WQ ? wof2 * hsum_wave : 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    pw[(size_t)mt * 4 * nbp] = WQ ? make_float4(wsc * acc[mt][0] + hs, wsc * acc[mt][1] + hs, wsc * acc[mt][2] + hs, wsc * acc[mt][3] + hs)
                                                  : make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
            }
        }
        // the output row of the PREVIOUS step goes out here, behind the polls: vector memory operations complete in
        // order, and a store queued in front of the poll loads would sit on the hand-off's critical path
        if (gate_wave && lane_on && step > t_begin)
        {
            const size_t fr = (size_t)(dir == 0 ? step - 1 : T - step);
            // (every store instruction of the step costs: with the planes the fp32 row is written only for the debug taps -- measured at
            // 32 lanes, per launch: one store 6.40 ms, two 6.56, three 6.85; the LAST row of a launch is always written, below)
            if (!plp || a.write_f32)
                outp[fr * ldo] = hlast; // lstm.cpp:163-164,170-171
            if (plp)
            {
                plp[fr * ldpl] = (unsigned short)(plast & 0xffffu);
                plp[plane_elems + fr * ldpl] = (unsigned short)(plast >> 16);
            }
        }
        // W_ih x + b_ih of this lane's unit and track (in the ring since at least one barrier ago)
        float4 p4 = p4s;
        if (SP == 1 && gate_wave && n < nbp)
            p4 = *reinterpret_cast<const float4 *>(ring + (size_t)((step & ring_mask) * nbp + n) * RING_PITCH + 16 * (4 * w + q));
        if (prof)
            c2 = clock64();
        __syncthreads();
        if (*abort_flag)
            return;
        if (prof)
            c3 = clock64();
        if (gate_wave && n < nbp)
        {
            __builtin_amdgcn_s_setprio(1); // the serial gate phase wins issue arbitration (measured: -1.5 % per segment pipelined)
            // the NDW k-range partials, summed in a fixed tree, two gates per (unswizzled) packed add
            float2v pa[8], pb[8];
#pragma unroll
            for (int ww = 0; ww < NDW; ++ww)
            {
                const float4 v4 = part[((size_t)((((step & 1) * 8 + ww) * MT + w) * 4) + q) * nbp + n];
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
            const float c_t = f_t * c + i_t * g_t; // lstm.cpp:154-156
            const float h = o_t * (PRECISE ? tanhf(c_t) : tanh_hw(c_t)); // lstm.cpp:157
            // h split in three bf16 terms (WQ: two fp16 terms of h * 2^14); the odd unit of the pair sits 16 lanes up in
            // this wave
            unsigned b1, b2, b3;
            if (WQ)
            {
                const float hs14 = h * HSCALE;
                const _Float16 h1 = (_Float16)hs14, h2 = (_Float16)(hs14 - (float)h1);
                b1 = __builtin_bit_cast(unsigned short, h1);
                b2 = __builtin_bit_cast(unsigned short, h2);
                b3 = 0u;
            }
            else
            {
                b1 = cvt_pk_bf16(h, 0.f) & 0xffffu;
                const float r1 = h - __uint_as_float(b1 << 16);
                b2 = cvt_pk_bf16(r1, 0.f) & 0xffffu;
                const float r2 = r1 - __uint_as_float(b2 << 16);
                b3 = cvt_pk_bf16(r2, 0.f) & 0xffffu;
            }
            const unsigned mine12 = b1 | (b2 << 16);
            // the partner's halves (lane ^ 16; only the even rows, which publish, use them): one v_permlane16_swap per value puts
            // row r + 1 into row r of the second result -- a VALU exchange instead of the LDS round trip of ds_swizzle
            const unsigned other12 = __builtin_amdgcn_permlane16_swap(mine12, mine12, false, false)[1];
            const unsigned other3 = WQ ? 0u : __builtin_amdgcn_permlane16_swap(b3, b3, false, false)[1];
            if (lane_on)
            {
                c = c_t;
                hlast = h;
                plast = mine12;
                if ((q & 1) == 0) // publish the pair (this unit, the next), tagged step + 1
                {
                    const uint4 gv = make_uint4(tag_hi | (unsigned)(step + 1), b1 | (other12 << 16), (mine12 >> 16) | (other12 & 0xffff0000u),
                                                WQ ? 0u : (b3 | (other3 << 16)));
                    granule_store16<FAST>(gran_rs, (int)(lstmb_granule_index(step & 1, chain, unit, n, HL, nbp) * 16), gv);
                }
            }
            __builtin_amdgcn_s_setprio(0);
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
    if (t_end > t_begin)
        row_sum_of_step(t_end - 1);
    if (gate_wave && lane_on) // lstm.cpp:160-161: the state carries into the next segment (and the next launch)
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
        a.state_out[st_h + unit] = hlast;
        a.state_out[st_c + unit] = c;
    }
    if (prof && l == 0)
    {
        for (int i = 0; i < 6; ++i)
            a.prof[(a.layer * 2 + pw_idx) * 8 + i] = (t_begin == 0 ? 0ull : a.prof[(a.layer * 2 + pw_idx) * 8 + i]) + pc[i];
        a.prof[(a.layer * 2 + pw_idx) * 8 + 6] = pc6;
        a.prof[(a.layer * 2 + pw_idx) * 8 + 7] = pc7;
    }
}

// grid = 8*S workgroups (1-D), plain launch; chunks of more than one step need the grid co-resident (census = 1)
template <int HL, bool WQ, bool PRECISE> __global__ __launch_bounds__(LSTM_THREADS) void lstm_batch_kernel(LstmBArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lstmb_smem[];
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
    // the launch serves the group of 16 lanes that starts at lane_base (contexts of more than 16 lanes that lstm_batch8.h does not
    // take -- hidden 128 / 256, fp32-resident W_hh -- run their groups one launch after the other: engine_lstm.h)
    const int group = a.lane_base / LSTMB_GROUP_TRACKS;
    if (s_ctl[2])
        lstmb_body<HL, WQ, true, PRECISE>(a, chain, slice, lstmb_smem, &s_ctl[3], group);
    else
        lstmb_body<HL, WQ, false, PRECISE>(a, chain, slice, lstmb_smem, &s_ctl[3], group);
}


// The row sums of the ONE row per direction that no later step multiplies with (the launch's last step: frame T - 1 of the forward
// chain, frame 0 of the backward chain): sum_k of the two fp16 planes of h_k * 2^14 -- formed here from the fp32 row exactly as the
// gate lanes / split_planes_kernel form them, so it does not matter who wrote the planes -- in a fixed tree over the lanes.  (The
// order of lstmb_body's all-ones tile is the matrix pipe's own; this row has its own fixed one, and every path runs this kernel for
// this row: a track's bits do not depend on the path.)  grid (lanes, targets x 2 dirs), 64 threads.
__global__ __launch_bounds__(64) void lstm_last_row_sum_kernel(LstmBArgs a, int ntargets)
{
    const int ln = blockIdx.x, target = a.tmap[blockIdx.y >> 1], dir = blockIdx.y & 1, l = threadIdx.x, HL = a.Hl;
    if (!((a.lane_mask >> ln) & 1ull) || (int)(blockIdx.y >> 1) >= ntargets || !a.rs_dir[target])
        return;
    const size_t fr = dir == 0 ? (size_t)a.T - 1 : 0, row = (size_t)ln * a.Tp + fr;
    const float *h = a.out[target] + (size_t)ln * a.out_stride + fr * a.ldo + a.col0 + dir * HL;
    float s = 0.f;
    for (int k = l; k < HL; k += 64) // lane l: units l, l + 64, ...; then the 64 lanes by xor-exchange
    {
        const float hs14 = h[k] * 16384.0f;
        const _Float16 h1 = (_Float16)hs14, h2 = (_Float16)(hs14 - (float)h1);
        s += (float)h1 + (float)h2;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if (l == 0)
        a.rs_dir[target][(size_t)dir * a.rs_rows + row] = s * (1.0f / 16384.0f);
}

} // namespace umx
