// wav.cpp -- minimal RIFF/WAVE reader + float32 writer (host side of include/umx_host.h).
// Stands in for libnyquist as used by the reference's load_audio / write_audio_file
// (src/dsp.cpp:18-101): decode to float, mono duplicated to both channels (dsp.cpp:52-60), only
// 44.1 kHz (dsp.cpp:27-33) and 1 or 2 channels (dsp.cpp:39-44); write 2-channel IEEE-float WAV
// (dsp.cpp:97-99 PCM_FLT).  libnyquist is an un-vendored submodule in the reference
// (.gitmodules:4-6), so its integer->float scaling is restated from its published macros
// (int16 / 32767.f, int24 / 8388608.f, int32 / 2147483648.f) and is not pinned by a test.
#include "../../include/umx_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace
{
void seterr(char *err, const std::string &m)
{
    if (err)
        snprintf(err, UMX_ERRLEN, "%s", m.c_str());
}
uint32_t u32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t u16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
} // namespace

extern "C" int umx_wav_load(const char *path, float **audio_out, int *n_frames_out, int *channels_in_file, char *err)
{
    if (!path || !audio_out || !n_frames_out)
    {
        seterr(err, "umx_wav_load: null argument");
        return UMX_ERR_ARG;
    }
    *audio_out = nullptr;
    *n_frames_out = 0;
    FILE *f = fopen(path, "rb");
    if (!f)
    {
        seterr(err, std::string("cannot open ") + path);
        return UMX_HOST_ERR_IO;
    }
    std::vector<unsigned char> b;
    {
        unsigned char chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof chunk, f)) > 0)
            b.insert(b.end(), chunk, chunk + n);
        fclose(f);
    }
    if (b.size() < 12 || memcmp(b.data(), "RIFF", 4) || memcmp(b.data() + 8, "WAVE", 4))
    {
        seterr(err, "not a RIFF/WAVE file");
        return UMX_HOST_ERR_AUDIO;
    }
    int fmt_tag = 0, channels = 0, rate = 0, bits = 0, block_align = 0;
    const unsigned char *data = nullptr;
    size_t data_len = 0;
    size_t pos = 12;
    while (pos + 8 <= b.size())
    {
        const unsigned char *ck = b.data() + pos;
        size_t len = u32(ck + 4);
        size_t body = pos + 8;
        if (!memcmp(ck, "fmt ", 4) && len >= 16 && body + 16 <= b.size())
        {
            fmt_tag = u16(b.data() + body);
            channels = u16(b.data() + body + 2);
            rate = (int)u32(b.data() + body + 4);
            block_align = u16(b.data() + body + 12);
            bits = u16(b.data() + body + 14);
            if (fmt_tag == 0xFFFE && len >= 26 && body + 26 <= b.size()) // WAVE_FORMAT_EXTENSIBLE
                fmt_tag = u16(b.data() + body + 24);
        }
        else if (!memcmp(ck, "data", 4))
        {
            data = b.data() + body;
            data_len = std::min(len, b.size() - body);
            break;
        }
        pos = body + len + (len & 1);
    }
    if (!data || !channels)
    {
        seterr(err, "WAVE file has no fmt/data chunk");
        return UMX_HOST_ERR_AUDIO;
    }
    if (channels_in_file)
        *channels_in_file = channels;
    if (rate != UMX_SAMPLE_RATE) // dsp.cpp:27-33
    {
        seterr(err, "[ERROR] umx.cpp only supports the following sample rate (Hz): 44100");
        return UMX_HOST_ERR_AUDIO;
    }
    if (channels != 1 && channels != 2) // dsp.cpp:39-44
    {
        seterr(err, "[ERROR] umx.cpp only supports mono and stereo audio");
        return UMX_HOST_ERR_AUDIO;
    }
    const int bps = bits / 8;
    if (!((fmt_tag == 1 && (bits == 16 || bits == 24 || bits == 32)) || (fmt_tag == 3 && bits == 32)) ||
        block_align != bps * channels)
    {
        seterr(err, "unsupported WAVE encoding (need PCM 16/24/32 or 32-bit float)");
        return UMX_HOST_ERR_AUDIO;
    }
    const size_t n = data_len / ((size_t)bps * channels);
    float *out = (float *)malloc(sizeof(float) * 2 * std::max<size_t>(n, 1));
    if (!out)
    {
        seterr(err, "out of memory");
        return UMX_HOST_ERR_IO;
    }
    auto sample = [&](size_t i) -> float {
        const unsigned char *p = data + i * bps;
        if (fmt_tag == 3)
        {
            float v;
            memcpy(&v, p, 4);
            return v;
        }
        if (bits == 16)
            return (float)(int16_t)u16(p) / 32767.f;
        if (bits == 24)
        {
            int32_t v = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16));
            if (v & 0x800000)
                v |= ~0xFFFFFF;
            return (float)v / 8388608.f;
        }
        return (float)(int32_t)u32(p) / 2147483648.f;
    };
    for (size_t i = 0; i < n; ++i)
    {
        if (channels == 1) // dsp.cpp:52-60
            out[2 * i] = out[2 * i + 1] = sample(i);
        else // dsp.cpp:62-69
        {
            out[2 * i] = sample(2 * i);
            out[2 * i + 1] = sample(2 * i + 1);
        }
    }
    *audio_out = out;
    *n_frames_out = (int)n;
    return UMX_OK;
}

extern "C" void umx_wav_free(float *audio) { free(audio); }

extern "C" int umx_wav_write_f32(const char *path, const float *audio, int n_frames, char *err)
{
    if (!path || !audio || n_frames < 0)
    {
        seterr(err, "umx_wav_write_f32: bad argument");
        return UMX_ERR_ARG;
    }
    FILE *f = fopen(path, "wb");
    if (!f)
    {
        seterr(err, std::string("cannot create ") + path);
        return UMX_HOST_ERR_IO;
    }
    const uint32_t data_bytes = (uint32_t)n_frames * 2u * 4u;
    unsigned char h[44];
    auto p32 = [&](int o, uint32_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; h[o + 2] = (v >> 16) & 255; h[o + 3] = (v >> 24) & 255; };
    auto p16 = [&](int o, uint16_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; };
    memcpy(h, "RIFF", 4);
    p32(4, 36 + data_bytes);
    memcpy(h + 8, "WAVEfmt ", 8);
    p32(16, 16);
    p16(20, 3); // WAVE_FORMAT_IEEE_FLOAT
    p16(22, 2);
    p32(24, UMX_SAMPLE_RATE);
    p32(28, UMX_SAMPLE_RATE * 2 * 4);
    p16(32, 8);
    p16(34, 32);
    memcpy(h + 36, "data", 4);
    p32(40, data_bytes);
    bool ok = fwrite(h, 1, 44, f) == 44 && fwrite(audio, 4, (size_t)n_frames * 2, f) == (size_t)n_frames * 2;
    ok = (fclose(f) == 0) && ok;
    if (!ok)
    {
        seterr(err, std::string("short write to ") + path);
        return UMX_HOST_ERR_IO;
    }
    return UMX_OK;
}
