// oracle/selftest.cpp -- drives the CPU restatement once through every entry point the parity tests use, for
// `make -C oracle sanitize` (AddressSanitizer + UBSan).  TEST INFRASTRUCTURE, like everything under oracle/.
// A small synthetic model (hidden 32) built through oracle_model_from_arrays; one short segment through
// oracle_umx_inference (with and without Wiener, with a skipped target), a ragged track through oracle_split_inference
// and oracle_shift_inference.  Exits non-zero if an output is not finite; the sanitizers abort on their own findings.
#include "umx_oracle.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static unsigned rng_state = 12345u;
static float urand() // [-1, 1)
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 8) / 8388608.0f - 1.0f;
}

static bool finite_all(const std::vector<float> &v)
{
    for (float x : v)
        if (!std::isfinite(x))
            return false;
    return true;
}

int main()
{
    const int hidden = 32;
    std::vector<std::vector<float>> store;
    std::vector<const float *> ptrs;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < UMXO_TENSORS_PER_TARGET; ++i)
        {
            const size_t n = oracle_tensor_numel(i, hidden);
            const std::string name = oracle_tensor_name(i);
            std::vector<float> a(n);
            for (size_t k = 0; k < n; ++k)
            {
                const float u = urand();
                if (name.find("running_var") != std::string::npos || name == "output_scale" ||
                    (name.find("bn") == 0 && name.find("weight") != std::string::npos))
                    a[k] = 1.0f + 0.5f * u;
                else if (name == "input_scale")
                    a[k] = 0.05f + 0.02f * u;
                else
                    a[k] = 0.1f * u;
            }
            store.push_back(std::move(a));
        }
    for (auto &a : store)
        ptrs.push_back(a.data());
    oracle_model *m = oracle_model_from_arrays(hidden, ptrs.data());
    if (!m)
        return std::printf("model_from_arrays failed\n"), 1;

    const int n_buf = 8192, n = 7000;
    std::vector<float> audio(2 * (size_t)n);
    for (float &x : audio)
        x = 0.5f * urand();
    std::vector<float> state(oracle_stream_state_floats(hidden), 0.f);
    std::vector<std::vector<float>> out(4, std::vector<float>(2 * (size_t)n));
    float *outp[4] = {out[0].data(), out[1].data(), out[2].data(), out[3].data()};
    int rc = 0;
    for (int flags : {0, 1, 0x200})
    {
        oracle_umx_inference(m, audio.data(), n, n_buf, state.data(), outp, flags, nullptr);
        for (auto &o : out)
            if (!finite_all(o))
                rc = 1;
    }
    const int length = 20000;
    std::vector<float> track(2 * (size_t)length);
    for (float &x : track)
        x = 0.5f * urand();
    std::vector<std::vector<float>> tout(4, std::vector<float>(2 * (size_t)length));
    float *toutp[4] = {tout[0].data(), tout[1].data(), tout[2].data(), tout[3].data()};
    oracle_split_inference(m, track.data(), length, n_buf, toutp, 0);
    for (auto &o : tout)
        if (!finite_all(o))
            rc = 1;
    oracle_shift_inference(m, track.data(), length, n_buf, 4033, toutp, 0);
    for (auto &o : tout)
        if (!finite_all(o))
            rc = 1;
    oracle_model_free(m);
    std::printf(rc ? "selftest: NON-FINITE OUTPUT\n" : "selftest: ok\n");
    return rc;
}
