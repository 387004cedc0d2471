// umx_cli.cpp -- `umx-cli <model file> <wav file> <out dir>`: the reference's CLI (umx.cpp:26-97)
// over the MI355X engine.  Loads the wav, loads the ggml weight file, creates the device context,
// runs shift_inference -> split_inference on the device (umx_hip_shift_inference: track resident in HBM,
// segments pipelined; UMX_CLI_PER_SEGMENT=1 selects the host drivers over umx_hip_infer_segment), writes
// target_{0..3}.wav (0 = bass, 1 = drums, 2 = other, 3 = vocals).  Exit code 1 on any failure,
// like the reference.  Extra knobs come from the environment only, so the 3 positionals stay:
//   UMX_DEVICE=<n>   UMX_NO_WIENER=1   UMX_SHIFT_OFFSET=<n>   UMX_LSTM_STEPWISE=1   UMX_CLI_PER_SEGMENT=1
//   UMX_WEIGHTS_RESIDENT=expanded   UMX_GEMM=f32
#include "../../include/umx_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <string>
#include <vector>

namespace
{
struct HipBackend
{
    umx_hip_ctx *ctx;
    unsigned flags;
};
int hip_segment(void *user, const float *audio, int n, float *const out[4])
{
    HipBackend *b = static_cast<HipBackend *>(user);
    int rc = umx_hip_infer_segment(b->ctx, audio, n, out, b->flags);
    if (rc)
        fprintf(stderr, "umx_hip_infer_segment: %s\n", umx_hip_last_error(b->ctx));
    return rc;
}
int hip_reset(void *user) { return umx_hip_stream_reset(static_cast<HipBackend *>(user)->ctx); }
void print_progress(float p, void *) { fprintf(stdout, "inference progress: %.1f %%\n", 100.f * p); }
int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
} // namespace

int main(int argc, const char **argv)
{
    if (argc != 4) // umx.cpp:28-33
    {
        fprintf(stderr, "Usage: %s <model file> <wav file> <out dir>\n", argv[0]);
        return 1;
    }
    const std::string model_file = argv[1], wav_file = argv[2], out_dir = argv[3];
    char err[UMX_ERRLEN] = "";
    printf("umx-cli (MI355X / gfx950) main driver program\n");

    float *audio = nullptr;
    int n = 0, ch = 0;
    if (umx_wav_load(wav_file.c_str(), &audio, &n, &ch, err)) // umx.cpp:56
    {
        fprintf(stderr, "%s\n", err);
        return 1;
    }
    printf("Input Samples: %d\nLength in seconds: %f\nNumber of channels: %d\n", n * ch, n / 44100.0, ch);

    const auto t0 = std::chrono::steady_clock::now();
    umx_model *model = nullptr;
    if (umx_model_load(model_file.c_str(), &model, err)) // umx.cpp:63-70
    {
        fprintf(stderr, "Error loading model: %s\n", err);
        return 1;
    }
    const auto t1 = std::chrono::steady_clock::now();
    printf("Loaded model (%d tensors, %6.2f MB) in %f s\n", umx_model_n_tensors(model),
           umx_model_data_bytes(model) / 1024.0 / 1024.0, std::chrono::duration<double>(t1 - t0).count());

    umx_hip_ctx *ctx = nullptr;
    if (umx_hip_create(&ctx, env_int("UMX_DEVICE", 0), umx_model_hidden(model), UMX_SEGMENT_SAMPLES,
                       umx_model_views(model), umx_model_n_tensors(model)))
    {
        fprintf(stderr, "umx_hip_create: %s\n", umx_hip_last_error(nullptr));
        return 1;
    }
    umx_model_free(model); // weights now live in HBM

    HipBackend hb{ctx, 0};
    if (env_int("UMX_NO_WIENER", 0))
        hb.flags |= UMX_FLAG_NO_WIENER;
    if (env_int("UMX_LSTM_STEPWISE", 0))
        hb.flags |= UMX_FLAG_LSTM_STEPWISE;
    umx_backend be{hip_segment, hip_reset, &hb};
    std::vector<float> stems[4];
    float *out[4];
    for (int t = 0; t < 4; ++t)
    {
        stems[t].resize((size_t)2 * n);
        out[t] = stems[t].data();
    }
    const auto t2 = std::chrono::steady_clock::now();
    const bool per_segment = env_int("UMX_CLI_PER_SEGMENT", 0) != 0;
    if (per_segment)
    {
        if (umx_shift_inference(&be, audio, n, UMX_SEGMENT_SAMPLES, env_int("UMX_SHIFT_OFFSET", -1), out,
                                print_progress, nullptr, err)) // umx.cpp:72-73
        {
            fprintf(stderr, "inference failed: %s\n", err);
            return 1;
        }
    }
    else if (umx_hip_shift_inference(ctx, audio, n, env_int("UMX_SHIFT_OFFSET", -1), out, hb.flags, print_progress,
                                     nullptr)) // umx.cpp:72-73
    {
        fprintf(stderr, "inference failed: %s\n", umx_hip_last_error(ctx));
        return 1;
    }
    const auto t3 = std::chrono::steady_clock::now();
    const double secs = std::chrono::duration<double>(t3 - t2).count();
    printf("Separated %.2f s of audio in %.3f s (%.1fx realtime, host buffers in/out, %s)\n", n / 44100.0, secs,
           n / 44100.0 / secs, per_segment ? "one segment at a time" : "track resident in HBM");

    std::error_code ec;
    std::filesystem::create_directories(out_dir, ec); // umx.cpp:84-86
    for (int t = 0; t < 4; ++t)                       // umx.cpp:75-96
    {
        const std::string p = (std::filesystem::path(out_dir) / ("target_" + std::to_string(t) + ".wav")).string();
        printf("Writing wav file %s\n", p.c_str());
        if (umx_wav_write_f32(p.c_str(), out[t], n, err))
        {
            fprintf(stderr, "%s\n", err);
            return 1;
        }
    }
    umx_wav_free(audio);
    umx_hip_destroy(ctx);
    return 0;
}
