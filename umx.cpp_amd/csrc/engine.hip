// engine.hip -- C-ABI implementation (include/umx_hip.h) of the gfx950 UMX segment engine.
// Orchestrates one segment exactly like umx_inference (inference.cpp:12-207):
//   stft -> |.| / crop+stack -> 4 x [fc1 bn tanh -> 3-layer BiLSTM -> fc2 bn relu -> fc3 bn scale
//   relu -> mask*mix] -> Wiener EM -> 4 x istft
// The four targets run inside the same launches; consecutive segments alternate between two pipeline
// slots (streams) so that their LSTM layers overlap as an exact wavefront (see struct Slot).  A context created
// for several tracks (umx_hip_create_tracks) runs one segment of each track per call: every stage is queued for
// all track lanes, and the LSTM recurrence of all lanes is ONE launch (lstm_batch.h).  Also here: the
// whole-track drivers (split / shift inference with the track resident in HBM), the phased form of a segment
// for the multi-GPU state-carry mode, weight residency (u8/u16 as stored, or expanded) and the GEMM flavour.
#include "../../include/umx_hip.h"
#include <chrono>
#include <dlfcn.h>

#include <algorithm>
#include <numeric>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "gemm_bf16x3.h"
#include "gemm_planes.h"
#include "gemm_planes_pp.h"
#include "gemm_planes_ps.h"
#include "lstm_kernels.h"
#include "lstm_batch.h"
#include "lstm_batch8.h"
#include "track_kernels.h"
#include "stft_kernels.h"
#include "wiener_kernels.h"
#include "wiener_istft.h"

using namespace umx;

// UMX_HIP_CHECK for the free functions of the C-ABI (the error goes to the context)
#define UMX_HIP_CHECK_CTX(ctx_, expr)                                                                                \
    do                                                                                                               \
    {                                                                                                                \
        hipError_t _e = (expr);                                                                                      \
        if (_e != hipSuccess)                                                                                        \
        {                                                                                                            \
            (ctx_)->set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                                    \
            return UMX_ERR_HIP;                                                                                      \
        }                                                                                                            \
    } while (0)

namespace
{
std::string g_create_error = "";

// A GEMM weight kept in HBM as stored in the ggml file (BASELINE config 5): u8 / u16 + (scale, offset);
// two parameter pairs because W_ih of a layer is two file tensors (forward rows, then reverse rows).
struct QMat
{
    void *q = nullptr;
    int type = 0; // GemmBType
    float s[2] = {1.f, 1.f}, o[2] = {0.f, 0.f};
};

// A GEMM weight as fp16 planes [nbp][N][K] (gemm_planes.h): u8 -> 1 exact plane of q - 128, u16 -> 2 planes whose sum is
// exactly q - 32896: fp16(q - 32896) and the remainder (an integer of at most 16), fp32 -> 2 split terms of w / s with s a
// power of two; (scale, offset + c scale) per file tensor (W_ih: two).
struct PMat
{
    unsigned short *p = nullptr;
    int nbp = 2;
    float s[2] = {1.f, 1.f}, o2[2] = {0.f, 0.f};
};

struct TargetBufs // weights of one target (shared by both pipeline slots)
{
    PMat fc1_p, ih_p[3], fc2_p, fc3_p; // gemm_planes.h (the default GEMM flavour)
    QMat fc1_q, ih_q[3], fc2_q, fc3_q; // used instead of the fp32 matrix when .q != nullptr
    unsigned short *fc1_bx = nullptr, *ih_bx[3] = {}, *fc2_bx = nullptr, *fc3_bx = nullptr; // bf16 planes [3][N][K] (gemm_bf16x3.h)
    float *fc1_w = nullptr, *in_scale = nullptr, *in_mean = nullptr, *bn1[4] = {};
    float *ih_w[3] = {}, *ih_b[3] = {};
    float *fc2_w = nullptr, *bn2[4] = {};
    float *fc3_w = nullptr, *bn3[4] = {}, *out_scale = nullptr, *out_mean = nullptr;
};

struct TargetAct // activations of one target in one pipeline slot
{
    float *cat = nullptr, *la = nullptr, *lb = nullptr, *P = nullptr, *a2 = nullptr,
          *mag = nullptr; // mag: fc3's MASK [2][T][MAGP]; x |X| = the target magnitude, formed by the consumers (gemm_common.h)
    // gemm_planes.h: every GEMM's A operand split once into two fp16 planes of the scaled row, its row sums and (for the
    // unbounded tensors) its per-row inverse scales
    unsigned short *xs_p = nullptr, *cat_p = nullptr, *la_p = nullptr, *lb_p = nullptr, *a2_p = nullptr;
    float *rs_xs = nullptr, *rs_catL = nullptr, *rs_catR = nullptr, *rs_la = nullptr, *rs_lb = nullptr, *rs_a2 = nullptr;
    float *rsc_xs = nullptr, *rsc_a2 = nullptr;
};

enum
{
    ST_STFT = 0,
    ST_FC1,
    ST_IH0,
    ST_LSTM0,
    ST_IH1,
    ST_LSTM1,
    ST_IH2,
    ST_LSTM2,
    ST_FC2,
    ST_FC3,
    ST_WIENER,
    ST_ISTFT,
    ST_OLA,
    ST_COUNT
};
enum { SP_XS = 0, SP_CATL, SP_LA, SP_LB, SP_CATR, SP_A2 }; // which A operand launch_split prepares
const char *kStageNames[ST_COUNT] = {"stft",  "fc1", "lstm_ih0", "lstm_rec0", "lstm_ih1", "lstm_rec1", "lstm_ih2",
                                     "lstm_rec2", "fc2", "fc3_mask", "wiener",  "istft",    "ola"};

// roctx ranges per stage (tracing): every stage marker below also opens a named range on the calling thread
// ("umx:<stage>": the HOST span in which the stage's kernels are queued; rocprofv3 --marker-trace shows them beside the
// kernel trace).  The marker library is used only if a profiler has ALREADY loaded it into the process (RTLD_NOLOAD) or
// UMX_ROCTX=1 asks for it: an installed ROCm alone does not make every stage marker a library call on the queuing path.
namespace
{
struct RoctxApi
{
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    char names[ST_COUNT][40];
    RoctxApi()
    {
        for (int i = 0; i < ST_COUNT; ++i)
            snprintf(names[i], sizeof names[i], "umx:%s", kStageNames[i]);
        const char *want = getenv("UMX_ROCTX");
        const bool force = want && atoi(want) != 0;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"})
            if (void *h = dlopen(lib, force ? (RTLD_NOW | RTLD_LOCAL) : (RTLD_NOW | RTLD_NOLOAD)))
            {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop)
                    return;
                push = nullptr;
                pop = nullptr;
            }
    }
};
thread_local bool t_stage_range_open = false;
// closes the calling thread's open stage range and opens `stage` (< 0: only closes; every exit of a call, error returns and
// umx_hip_segment_discard included, ends with stage_range(-1))
inline void stage_range(int stage)
{
    static const RoctxApi api;
    if (!api.push)
        return;
    if (t_stage_range_open)
        api.pop();
    t_stage_range_open = stage >= 0 && stage < ST_COUNT;
    if (t_stage_range_open)
        api.push(api.names[stage]);
}
struct StageRangeCloser // closes whatever stage range is open when a queuing function returns, on every path
{
    ~StageRangeCloser() { stage_range(-1); }
};
} // namespace

// One track lane of a pipeline slot = the activations of one in-flight segment of one track.
struct Lane
{
    TargetAct ta[4];
    float2 *spec = nullptr, *y = nullptr, *frames = nullptr;
    float *x = nullptr, *wpart = nullptr, *Rc = nullptr;
    unsigned *maxabs = nullptr;
};

// One pipeline slot = everything one in-flight segment (of every track lane) needs.  Two slots on two streams let
// segment s+1 run its STFT/GEMMs and LSTM layer l while segment s is in layer l+1 (the exact wavefront of
// R_l(s+1) waits only for R_l(s) through an event); the single-track LSTM kernels are latency-bound
// and use half of each CU's wave slots, so two of them co-reside and fill each other's hand-off gaps.
struct Slot
{
    hipStream_t stream = nullptr;
    Lane lane[LSTMB_MAX_TRACKS];
    float *hbuf = nullptr;
    bool lstm_wrote_planes[3] = {false, false, false}; // this call's recurrence of layer l wrote the next GEMM's A planes (run_lstm_layer_batched)
    bool lstm_rows_f32[3] = {true, true, true};        // ... and its fp32 output rows as well (else only the launch's last row: the taps of that layer are unavailable)
    unsigned *status = nullptr, *lsync = nullptr;
    unsigned long long *lprof = nullptr;
    hipEvent_t ev[ST_COUNT + 1] = {};
    hipEvent_t evk[ST_COUNT] = {}; // GEMM stages of plane contexts: recorded between the stage's split kernel and its GEMM kernel
    bool evk_set[ST_COUNT] = {};
    hipEvent_t rec_done[3] = {}; // LSTM layer l of this slot's segment has finished (state updated)
    // host-pointer entry points (see umx_hip_infer_batch_async): the download of call k's stems is queued on the OTHER slot's
    // stream, behind call k + 1's kernels
    hipEvent_t k_done = nullptr;   // this slot's stems are complete (recorded on its own stream)
    hipEvent_t out_free = nullptr; // their download has finished (recorded on the other slot's stream): overlap-add may overwrite them
    bool out_free_valid = false;
    bool have_times = false, last_persistent = false, used = false;
};
} // namespace

// Admission of persistent LSTM grids, process-wide and per device.  A persistent launch needs ALL its workgroups
// co-resident (they exchange granules), so the grids that can be resident TOGETHER must fit the device together.
// Grids queued on one stream run one after the other, so a stream contributes at most its largest queued grid; what has
// to fit is the sum over streams.  A launch that would not fit is made to wait -- on the device, through a stream-wait
// on another stream's newest admitted grid (which, streams being in order, is a wait for all of that stream's grids),
// never by blocking the host -- for the streams whose queued grids are oldest, until it fits.  The streams it waits for
// hold only grids admitted earlier, i.e. work that does not depend on the new launch: no cycle.
// (Rounds 1-2 summed over all queued grids instead of over streams: with the host running ahead of the device, a launch then
// waited for every earlier grid but the most recent one, and the two-slot wavefront of a single-track context overlapped
// only L2(s) with L0(s+1) -- one pair per segment instead of every layer.  Found in round 3.)
// Units are half CUs: a single-track workgroup (two fit a CU) counts 1, a batched one (one per CU) counts 2.
namespace
{
struct LstmGate
{
    std::mutex m;
    struct Grid
    {
        hipEvent_t done;
        int units;
        hipStream_t stream;
        unsigned long long seq;
    };
    std::vector<Grid> inflight; // in admission order
    std::vector<hipEvent_t> pool;
    unsigned long long seq = 0;
    int reserved = 0; // half CUs kept free for kernels that are not this engine's (RCCL send / recv: umx_hip_gate_reserve)
    std::vector<int> reservations; // outstanding requests in CUs: `reserved` is twice their sum
};
LstmGate g_gate[16];

// admit + launch + record under the gate's lock (an event that has not been recorded yet would read as complete)
template <class Launch> hipError_t lstm_gate_launch(int device, hipStream_t st, int units, int capacity_units, Launch launch)
{
    LstmGate &g = g_gate[device & 15];
    std::lock_guard<std::mutex> lock(g.m);
    capacity_units -= std::min(g.reserved, capacity_units / 2);
    for (size_t i = 0; i < g.inflight.size();)
        if (hipEventQuery(g.inflight[i].done) == hipSuccess)
        {
            g.pool.push_back(g.inflight[i].done);
            g.inflight.erase(g.inflight.begin() + i);
        }
        else
            ++i;
    (void)hipGetLastError(); // hipEventQuery reports "not ready" as an error
    // per other stream: its largest queued grid, the admission number of its oldest one, and its newest grid's event
    struct PerStream
    {
        hipStream_t stream;
        int units;
        unsigned long long oldest;
        hipEvent_t newest;
    };
    std::vector<PerStream> others;
    for (const LstmGate::Grid &gr : g.inflight)
    {
        if (gr.stream == st)
            continue;
        PerStream *ps = nullptr;
        for (PerStream &o : others)
            if (o.stream == gr.stream)
                ps = &o;
        if (!ps)
        {
            others.push_back({gr.stream, 0, gr.seq, gr.done});
            ps = &others.back();
        }
        ps->units = std::max(ps->units, gr.units);
        ps->newest = gr.done; // admission order: the last one seen is the newest
    }
    int used = 0;
    for (const PerStream &o : others)
        used += o.units;
    while (used + units > capacity_units && !others.empty())
    {
        size_t k = 0;
        for (size_t i = 1; i < others.size(); ++i)
            if (others[i].oldest < others[k].oldest)
                k = i;
        (void)hipStreamWaitEvent(st, others[k].newest, 0); // every grid of that stream has left the device when we start
        used -= others[k].units;
        others.erase(others.begin() + k);
    }
    hipEvent_t ev = nullptr;
    if (!g.pool.empty())
    {
        ev = g.pool.back();
        g.pool.pop_back();
    }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
        ev = nullptr;
    const hipError_t e = launch();
    if (ev && e == hipSuccess && hipEventRecord(ev, st) == hipSuccess)
        g.inflight.push_back({ev, units, st, g.seq++});
    else if (ev)
        g.pool.push_back(ev);
    return e;
}
} // namespace

// kernel instantiation pickers
static const void *lstm_persistent_fn(int kpw, bool precise)
{
    return precise ? (kpw == 8    ? reinterpret_cast<const void *>(lstm_persistent_kernel<8, true>)
                      : kpw == 16 ? reinterpret_cast<const void *>(lstm_persistent_kernel<16, true>)
                      : kpw == 32 ? reinterpret_cast<const void *>(lstm_persistent_kernel<32, true>)
                                  : reinterpret_cast<const void *>(lstm_persistent_kernel<64, true>))
                   : (kpw == 8    ? reinterpret_cast<const void *>(lstm_persistent_kernel<8, false>)
                      : kpw == 16 ? reinterpret_cast<const void *>(lstm_persistent_kernel<16, false>)
                      : kpw == 32 ? reinterpret_cast<const void *>(lstm_persistent_kernel<32, false>)
                                  : reinterpret_cast<const void *>(lstm_persistent_kernel<64, false>));
}
template <int HL> static const void *lstm_batch_fn_hl(bool wq, bool precise)
{
    return wq ? (precise ? reinterpret_cast<const void *>(lstm_batch_kernel<HL, true, true>)
                         : reinterpret_cast<const void *>(lstm_batch_kernel<HL, true, false>))
              : (precise ? reinterpret_cast<const void *>(lstm_batch_kernel<HL, false, true>)
                         : reinterpret_cast<const void *>(lstm_batch_kernel<HL, false, false>));
}
#ifndef UMX_FUSE_LSTM_PLANES
#define UMX_FUSE_LSTM_PLANES 1
#endif
// octets of 8 lanes x column shards of 64 units (lstm_batch8.h): LSTM hidden 512 (UMX-L) and 256 (umxhq), u8-resident W_hh
template <int HL> static const void *lstm_batch8_fn_hl(bool precise, int no)
{
    if (no == 2)
        return precise ? reinterpret_cast<const void *>(lstm_batch8_kernel<HL, true, 2>) : reinterpret_cast<const void *>(lstm_batch8_kernel<HL, false, 2>);
    return precise ? reinterpret_cast<const void *>(lstm_batch8_kernel<HL, true, 1>) : reinterpret_cast<const void *>(lstm_batch8_kernel<HL, false, 1>);
}
static const void *lstm_batch8_fn(int Hl, bool precise, int no = 1) // no = 2: two octets per workgroup in turn (hidden 1024: 33 .. 64 lanes in one launch)
{
    return Hl == 512 ? lstm_batch8_fn_hl<512>(precise, no) : Hl == 256 ? lstm_batch8_fn_hl<256>(precise, no) : nullptr;
}
static const void *lstm_batch_fn(int Hl, bool wq, bool precise)
{
    switch (Hl)
    {
    case 64: return lstm_batch_fn_hl<64>(wq, precise);
    case 128: return lstm_batch_fn_hl<128>(wq, precise);
    case 256: return lstm_batch_fn_hl<256>(wq, precise);
    case 512: return lstm_batch_fn_hl<512>(wq, precise);
    default: return nullptr;
    }
}

struct umx_hip_ctx
{
    int device = 0, H = 0, Hl = 0, S = 0, N = 0, T = 0, Tp = 0, nbatch = 0;
    static constexpr int Mpad = 256; // slack rows behind the last lane: a launch's M is rounded up to the 256-row tile
    std::string err;
    std::vector<void *> allocs;
    TargetBufs tb[4];
    float *whh[3] = {}, *bhh[3] = {};
    float *window = nullptr, *nw = nullptr;
    float2 *tw1 = nullptr, *tw2 = nullptr;
    float *tap_tmp = nullptr;                    // umx_hip_read_tap: scratch of the computed taps
    float *audio_in = nullptr, *out_dev[4] = {}; // device staging of the phased (multi-GPU carry) entry points
    static constexpr int kMaxSlots = 3;
    float *stage_in[kMaxSlots] = {}, *stage_out[kMaxSlots][4 * LSTMB_MAX_TRACKS] = {}; // per pipeline slot: device staging of the host-pointer
    int ensure_staging();                                              // entry points, [lane] / [lane][4]; allocated on first use
    hipEvent_t order_ev = nullptr;
    struct DeferredDownload // the stems of a host-pointer call, in its slot's staging buffers
    {
        bool valid = false;
        int si = 0, nb = 0, n[LSTMB_MAX_TRACKS] = {};
        float *host[4 * LSTMB_MAX_TRACKS] = {};
    };
    int queue_download(const DeferredDownload &d, hipStream_t on); // D2H copies of d onto stream `on`, then out_free of its slot
    hipStream_t copy_stream = nullptr; // downloads of the host-pointer calls (created with the staging buffers)
    float *state = nullptr, *state_alt = nullptr; // state_alt: second copy for the per-step driver of the batched recurrence
    Slot slot[kMaxSlots];
    int nslots = 2; // pipeline slots: 3 for single-track contexts (see init), 2 for track-batched ones
    int next_slot() const { return (int)(nseg % nslots); }
    void clear_used()
    {
        for (int si = 0; si < kMaxSlots; ++si)
            slot[si].used = false;
    }
    int B = 1;                // track lanes (umx_hip_create_tracks); every lane has its own streaming LSTM state
    bool lstm_batched = false; // LSTM recurrence on the matrix cores for all lanes at once (lstm_batch.h); fixed at
                               // create so that a track's bits never depend on how many lanes a call uses
    unsigned tag_epoch = 0;   // persistent LSTM launches so far (granule tags are unique per launch)
    unsigned next_tag_base()
    {
        // 4096 tags per launch (T + 1 <= 4096 is checked at create); wraps after ~1M launches, where one
        // stale line from exactly 2^20 launches ago would have to survive in an L2 -- every buffer is
        // zeroed at that point anyway (see run_lstm_layer)
        tag_epoch = (tag_epoch + 1) & 0xFFFFF;
        return tag_epoch << 12;
    }
    int cur = 0;              // slot of the most recently queued segment
    long long nseg = 0;       // segments queued since creation
    size_t lsync_words = 0;
    bool persistent_ok = true;
    int n_cus = 256;
    // what umx_hip_sync needs to redo the calls queued since the last sync after a persistent-kernel timeout
    struct PendingCall
    {
        int nb;
        const float *audio[LSTMB_MAX_TRACKS];
        int n[LSTMB_MAX_TRACKS];
        float *out[4 * LSTMB_MAX_TRACKS];
        float *host_out[4 * LSTMB_MAX_TRACKS]; // host-pointer entry points: where the stems are copied afterwards
        const float *host_audio[LSTMB_MAX_TRACKS]; // ... and where the audio came from: the two staging buffers have been
                                                   // reused by later calls, so a replay uploads it again
        unsigned flags;
    };
    static constexpr int kBackupCalls = 8;
    std::vector<PendingCall> pending; // at most kBackupCalls entries
    bool pending_lost = false;        // calls were queued that cannot be replayed: more than kBackupCalls since the last sync,
                                      // or the caller was released from the buffer contract (umx_hip_order_before)
    float *backup = nullptr; // [kBackupCalls][3 layers][B * state_floats]: the stream state right before each layer launch
    bool no_recovery = false, recovering = false;
    int recover();
    int lstm_poll_delay = 0;         // 0 = the kernel's default (LSTM_POLL_DELAY)
    int env_gemm_ps = -1;            // UMX_GEMM_PS: bit per GemmMode, which 256 x 256 launches take the persistent kernel (gemm_planes_ps.h); < 0: all
    int env_gemm_pp = -1;            // UMX_GEMM_PP: bit mask of the GEMMs that take the ping-pong kernel; < 0: all.  Both are read ONCE, when the context is created: a test makes a context per mode
    const char *gemm_kernel_last[4] = {"none", "none", "none", "none"}; // per GemmMode (umx_hip_gemm_kernel_name)
    const char *lstm_kernel_last = "none"; // the recurrence kernel of the last layer launch (umx_hip_lstm_kernel_name)
    int lstm_threads = LSTM_THREADS; // 512 (two workgroups per CU fit) or 576 (dedicated gate wave)
    int lstm_capacity = 0;           // workgroups of the persistent LSTM kernel that can be co-resident
    unsigned last_flags = 0;
    hipStream_t stream = nullptr; // = slot[0].stream (H2D/D2H of the host-pointer entry point)

    void set_error(const std::string &s) { err = s; }

    template <class T_> int dalloc(T_ **p, size_t count, bool zero = true)
    {
        void *q = nullptr;
        UMX_HIP_CHECK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T_)));
        allocs.push_back(q);
        if (zero)
            UMX_HIP_CHECK(hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T_)));
        *p = reinterpret_cast<T_ *>(q);
        return UMX_OK;
    }
    template <class T_> int upload(T_ **p, const std::vector<T_> &h)
    {
        int rc = dalloc(p, h.size(), false);
        if (rc)
            return rc;
        UMX_HIP_CHECK(hipMemcpy(*p, h.data(), h.size() * sizeof(T_), hipMemcpyHostToDevice));
        return UMX_OK;
    }
    int init(int device_, int hidden, int segment_samples, const umx_tensor_view *tensors, int n_tensors,
             unsigned create_flags, int n_tracks);
    size_t weight_bytes = 0;      // HBM held by model tensors (the config-5 figure of merit)
    bool gemm_bf16x3 = false;     // dense stack on the bf16 matrix cores, three-term split (gemm_bf16x3.h)
    bool wiener_fused = true;     // wiener_istft.h: gains + filter + inverse STFT frame in one kernel
    bool gemm_planes = false;     // ... with both operands pre-split / re-encoded as bf16 planes and LDS-DMA staging (gemm_planes.h)
    void launch_split(Lane &ln, int nl, hipStream_t st, int which, const int *active, int nact);
    void launch_gemm_planes(Lane &ln, int nl, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg);
    // one GEMM stage for the track lanes with audio: the plane GEMMs take every run of consecutive lanes in one launch
    void launch_gemm_lanes(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, int mode, int layer, const int *active,
                           int nact, bool dbg);
    bool u8_dequant = false;      // UMX_CREATE_U8_DEQUANT: u8 weights dequantised per element (model.cpp:610-616) before they
                                  // are multiplied, instead of exact bf16 integers with the affine map applied to the sum
    unsigned char *whh_q[3] = {}; // u8-resident W_hh (create flag), same layout as whh[]
    float whh_s[3][8] = {}, whh_o[3][8] = {};
    int infer_device(const float *audio_dev, int n, float *const out[4], unsigned flags);
    // one segment of each of `nb` track lanes (lane i = track i of this context; audio[i] == nullptr: lane idle)
    int infer_batch(int nb, const float *const *audio_dev, const int *n, float *const *out /* [nb][4] */, unsigned flags);
    int lstm_batch_capacity = 0; // co-resident workgroups of the batched LSTM kernel
    bool lstm_batch8_ok = false; // lstm_batch8_kernel (octets of 8 lanes, lstm_batch8.h) serves this context's batched launches
    int lstm8_poll_delay = 0;    // UMX_LSTM8_POLL_DELAY (read at create)
    bool env_lstm8_paired = true; // UMX_LSTM8_PAIRED=0 (read at create): 33 .. 64 lanes as two launches of 32 instead of two octets per workgroup in turn
    int env_lstm8_min = 1;       // UMX_LSTM8_MIN_LANES (read at create): contexts of at least this many lanes use it (99: none)
    bool lstm_rowsums = false;       // the batched recurrence hands the consuming plane GEMM the row sums of its output (lstm_batch.h, LstmBArgs::rs_dir)
    bool lstm_writes_planes = false; // ... and writes that GEMM's A planes itself
    size_t state_floats() const { return (size_t)4 * 12 * Hl; }
    // phased form of one segment (exact multi-GPU carry): front | layer 0 | layer 1 | layer 2 | back
    // whole track on the device (split_inference / shift_inference, umx.cpp:99-295)
    int track(const float *audio_host, int length, int shift_offset, float *const out_host[4], unsigned flags,
              void (*progress)(float, void *), void *progress_user);
    int tracks(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host, unsigned flags,
               void (*progress)(float, void *), void *progress_user);
    int tracks_once(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host,
                    unsigned flags, void (*progress)(float, void *), void *progress_user);
    struct TrackBufs // whole-track driver, per track lane: the (shifted) track, 4 stem accumulators, weight sum, 2 x 4 segment stems
    {
        float *in = nullptr, *out[4] = {}, *sumw = nullptr, *seg[kMaxSlots][4] = {};
        size_t cap = 0;
    };
    std::vector<TrackBufs> trk;
    hipEvent_t trk_acc_ev[kMaxSlots] = {};
    int phase_begin(const float *audio_host, int n, unsigned flags);
    int phase_begin_device(const float *audio_dev, int n, unsigned flags);
    int phase_end_device(float *const out_dev_[4]);
    const float *ph_audio = nullptr;
    int phase_layer(int layer);
    int phase_end(float *const out_host[4]);
    int ph_next = -1; // -1: no phased segment open; 0..2: next LSTM layer; 3: back stage pending
    int ph_n = 0;
    unsigned ph_flags = 0;
    int run_lstm_layer(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask);
    int run_lstm_layer_batched(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask);
    int sync_all();
    void launch_gemm(Lane &ln, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg);
    int stage_front(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, const int *n, const int *active, int nact);
    // elements between the streaming buffers of consecutive track lanes of a slot (the kernels take lane 0's pointers)
    WienerStrides lane_strides() const
    {
        WienerStrides ls;
        const size_t nchunk = (size_t)(T + WIENER_CHUNK - 1) / WIENER_CHUNK;
        ls.spec = (size_t)2 * T * NBINS;
        ls.mag = (size_t)2 * T * MAGP;
        ls.part = std::max((size_t)4 * nbatch * NBINS * 9, nchunk * 4 * 5 * NBINS);
        ls.rc = (size_t)4 * NBINS * 4;
        ls.frames = (size_t)4 * T * NFFT;
        ls.y = (size_t)4 * 2 * T * NBINS;
        return ls;
    }
    static LaneSet lane_set(int nb, const float *const *audio_dev) // the active lanes of a call
    {
        LaneSet l;
        l.count = 0;
        for (int ln = 0; ln < nb; ++ln)
            if (audio_dev[ln])
                l.id[l.count++] = (unsigned char)ln;
        return l;
    }
    int stage_back(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n,
                   unsigned flags, const int *active, int nact);
    // its two halves: fc2 + fc3 -> the target magnitudes of the active targets | Wiener (or mixture phase), inverse STFT,
    // overlap-add from the magnitudes of ALL four targets (zero_skipped: a skipped target counts as silence; false when
    // its magnitudes were put there by someone else -- the target-sharded multi-GPU driver)
    int stage_masks(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, unsigned flags, const int *active, int nact);
    int stage_finish(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n, unsigned flags,
                     bool zero_skipped);
    int phase_masks();
    int phase_finish_device(float *const out_dev_[4]);
    static void active_list(unsigned flags, int *active, int &nact)
    {
        nact = 0;
        for (int tg = 0; tg < 4; ++tg)
            if (!(flags & UMX_FLAG_SKIP_TARGET(tg)))
                active[nact++] = tg;
    }
};

#include "engine_init.h"
#include "engine_lstm.h"
#include "engine_stages.h"
#include "engine_tracks.h"
#include "engine_phased.h"
#include "engine_cabi.h"
