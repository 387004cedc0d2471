// umx_batch.cpp -- `umx-batch <model file> <out dir> <wav file> [<wav file> ...]`: the reference's CLI (umx.cpp:26-97) for
// up to 16 files at a time.  The reference is one track per process; on the GPU the LSTM recurrence of one track is
// bound by hand-off latency, not arithmetic, so the engine runs several tracks as track lanes of ONE context
// (umx_hip_create_tracks / umx_hip_separate_tracks): per file the same shift_inference -> split_inference, the same
// four stems, written to <out dir>/<wav stem>/target_{0..3}.wav.  More than 16 files are taken 16 at a time.
// Environment: UMX_DEVICE, UMX_NO_WIENER, UMX_SHIFT_OFFSET (as umx-cli).
#include "../../include/umx_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <set>
#include <string>
#include <vector>

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

int main(int argc, const char **argv)
{
    if (argc < 4)
    {
        fprintf(stderr, "Usage: %s <model file> <out dir> <wav file> [<wav file> ...]\n", argv[0]);
        return 1;
    }
    const std::string model_file = argv[1], out_dir = argv[2];
    const int nfiles = argc - 3;
    int lanes = std::min(nfiles, UMX_MAX_TRACKS);
    char err[UMX_ERRLEN] = "";
    umx_model *model = nullptr;
    if (umx_model_load(model_file.c_str(), &model, err)) // umx.cpp:63-70
    {
        fprintf(stderr, "Error loading model: %s\n", err);
        return 1;
    }
    umx_hip_ctx *ctx = nullptr;
    int rc = umx_hip_create_tracks(&ctx, env_int("UMX_DEVICE", 0), umx_model_hidden(model), UMX_SEGMENT_SAMPLES, umx_model_views(model),
                                   umx_model_n_tensors(model), 0, lanes);
    if (rc && lanes > 16) // more than 16 lanes need the u8-resident W_hh of a quantised model: an f32 model gets 16
    {
        lanes = 16;
        rc = umx_hip_create_tracks(&ctx, env_int("UMX_DEVICE", 0), umx_model_hidden(model), UMX_SEGMENT_SAMPLES, umx_model_views(model),
                                   umx_model_n_tensors(model), 0, lanes);
    }
    if (rc)
    {
        fprintf(stderr, "umx_hip_create_tracks: %s\n", umx_hip_last_error(nullptr));
        return 1;
    }
    umx_model_free(model);
    const unsigned flags = env_int("UMX_NO_WIENER", 0) ? UMX_FLAG_NO_WIENER : 0;
    double audio_secs = 0, wall = 0;
    const int first_rand_shift = UMX_REFERENCE_SHIFT; // glibc's first unseeded rand() % 22050 (not rand() here: umx_hip.h)
    std::set<std::string> used_names; // output directories are named after the file's stem: a/x.wav and b/x.wav must not collide
    for (int f0 = 0; f0 < nfiles; f0 += lanes)
    {
        const int nb = std::min(lanes, nfiles - f0);
        std::vector<float *> audio(nb, nullptr);
        // umx.cpp:115 draws rand() % 22050 once per PROCESS, and the reference never seeds rand(): every run of its CLI
        // shifts by the same 4033 samples.  Every file here gets that value too (not a fresh draw per batch of lanes),
        // so a file's stems do not depend on where in the argument list it stands; UMX_SHIFT_OFFSET overrides it.
        std::vector<int> n(nb, 0), shift(nb, env_int("UMX_SHIFT_OFFSET", -1) < 0 ? first_rand_shift : env_int("UMX_SHIFT_OFFSET", -1));
        std::vector<std::vector<float>> stems(4 * nb);
        std::vector<float *> out(4 * nb);
        std::vector<const float *> in(nb);
        for (int i = 0; i < nb; ++i)
        {
            int ch = 0;
            if (umx_wav_load(argv[3 + f0 + i], &audio[i], &n[i], &ch, err)) // umx.cpp:56
            {
                fprintf(stderr, "%s: %s\n", argv[3 + f0 + i], err);
                return 1;
            }
            in[i] = audio[i];
            audio_secs += n[i] / 44100.0;
            for (int t = 0; t < 4; ++t)
            {
                stems[4 * i + t].resize((size_t)2 * n[i]);
                out[4 * i + t] = stems[4 * i + t].data();
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (umx_hip_separate_tracks(ctx, nb, in.data(), n.data(), shift.data(), out.data(), flags, nullptr, nullptr))
        {
            fprintf(stderr, "inference failed: %s\n", umx_hip_last_error(ctx));
            return 1;
        }
        wall += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < nb; ++i)
        {
            std::string name = std::filesystem::path(argv[3 + f0 + i]).stem().string();
            for (int k = 2; !used_names.insert(name).second; ++k)
                name = std::filesystem::path(argv[3 + f0 + i]).stem().string() + "_" + std::to_string(k);
            const std::filesystem::path dir = std::filesystem::path(out_dir) / name;
            std::error_code ec;
            std::filesystem::create_directories(dir, ec);
            for (int t = 0; t < 4; ++t) // umx.cpp:75-96
            {
                const std::string p = (dir / ("target_" + std::to_string(t) + ".wav")).string();
                if (umx_wav_write_f32(p.c_str(), out[4 * i + t], n[i], err))
                {
                    fprintf(stderr, "%s\n", err);
                    return 1;
                }
            }
            umx_wav_free(audio[i]);
        }
    }
    printf("Separated %d file(s), %.2f s of audio in %.3f s (%.1fx realtime, host buffers in/out, %d track lanes)\n", nfiles, audio_secs, wall,
           audio_secs / wall, lanes);
    umx_hip_destroy(ctx);
    return 0;
}
