// model.cpp -- ggml-style UMX weight-file loader (host side of include/umx_host.h).
// Behaviour follows the reference's load_umx_model (src/model.cpp:42-574): gzip -> magic 0x756d7867
// -> hidden size -> records {scale, offset, n_dims, name_len, ne[], name, data} until EOF, name
// dispatch with a per-name expected shape, u16 for fc2/fc3/bn2/bn3 and u8 otherwise, target index
// advanced by bn3.running_var.  Unlike the reference the file is inflated in memory, tensors stay
// quantised (the device engine dequantises), and every failure is a status code + message.
#include "../../include/umx_host.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <zlib.h>

namespace
{
const int kCrop = 1487, kBins = 2049; // model.cpp:143-151

struct Rec
{
    std::string name;
    int target, dtype, n_dims, ne[2];
    float scale, offset;
    size_t data_off, nbytes;
};

void seterr(char *err, const std::string &m)
{
    if (err)
        snprintf(err, UMX_ERRLEN, "%s", m.c_str());
}

// expected ColMajor (ne0, ne1) of every tensor the loader knows: model.cpp:139-186
bool expected_shape(const std::string &n, int H, int *ne0, int *ne1)
{
    const int G = 2 * H, Hl = H / 2;
    *ne1 = 1;
    if (n == "input_mean" || n == "input_scale")
        *ne0 = kCrop;
    else if (n == "output_mean" || n == "output_scale")
        *ne0 = kBins;
    else if (n == "fc1.weight")
        *ne0 = 2 * kCrop, *ne1 = H;
    else if (n == "fc2.weight")
        *ne0 = G, *ne1 = H;
    else if (n == "fc3.weight")
        *ne0 = H, *ne1 = 2 * kBins;
    else if (n.rfind("bn1.", 0) == 0 || n.rfind("bn2.", 0) == 0)
        *ne0 = H;
    else if (n.rfind("bn3.", 0) == 0)
        *ne0 = 2 * kBins;
    else if (n.rfind("lstm.weight_ih_l", 0) == 0)
        *ne0 = H, *ne1 = G;
    else if (n.rfind("lstm.weight_hh_l", 0) == 0)
        *ne0 = Hl, *ne1 = G;
    else if (n.rfind("lstm.bias_ih_l", 0) == 0 || n.rfind("lstm.bias_hh_l", 0) == 0)
        *ne0 = G;
    else
        return false;
    if (n.rfind("bn", 0) == 0)
    {
        const std::string f = n.substr(4);
        if (f != "weight" && f != "bias" && f != "running_mean" && f != "running_var")
            return false;
    }
    if (n.rfind("lstm.", 0) == 0)
    {
        // lstm.<kind>_l<0..2>[_reverse]
        size_t p = n.find("_l");
        std::string tail = n.substr(p + 2);
        if (tail != "0" && tail != "1" && tail != "2" && tail != "0_reverse" && tail != "1_reverse" &&
            tail != "2_reverse")
            return false;
    }
    return true;
}

bool is_u16(const std::string &n) // convert-umx-pth-to-ggml.py:146 / model.cpp name dispatch
{
    return n.find("bn2") != std::string::npos || n.find("bn3") != std::string::npos ||
           n.find("fc2") != std::string::npos || n.find("fc3") != std::string::npos;
}
} // namespace

struct umx_model
{
    int hidden = 0;
    std::vector<unsigned char> blob;
    std::vector<Rec> recs;
    std::vector<umx_tensor_view> views;
    size_t data_bytes = 0;
    float progress = 0.f;
};

extern "C" int umx_model_load(const char *path, umx_model **out, char *err)
{
    if (!path || !out)
    {
        seterr(err, "umx_model_load: null argument");
        return UMX_ERR_ARG;
    }
    *out = nullptr;
    gzFile gz = gzopen(path, "rb"); // model.cpp:58 (gzread also passes an un-gzipped file through)
    if (!gz)
    {
        seterr(err, std::string("failed to open ") + path);
        return UMX_HOST_ERR_IO;
    }
    umx_model *m = new umx_model;
    {
        std::vector<unsigned char> chunk(1 << 20);
        int n;
        while ((n = gzread(gz, chunk.data(), (unsigned)chunk.size())) > 0)
            m->blob.insert(m->blob.end(), chunk.begin(), chunk.begin() + n);
        int zerr = 0;
        gzerror(gz, &zerr);
        gzclose(gz);
        if (n < 0 || (zerr != Z_OK && zerr != Z_STREAM_END))
        {
            delete m;
            seterr(err, "gzip stream is corrupt");
            return UMX_HOST_ERR_FORMAT;
        }
    }
    m->progress = 0.1f; // model.cpp:66
    const std::vector<unsigned char> &b = m->blob;
    size_t pos = 0;
    auto rd = [&](void *dst, size_t n) -> bool {
        if (pos + n > b.size())
            return false;
        memcpy(dst, b.data() + pos, n);
        pos += n;
        return true;
    };
    auto fail = [&](int code, const std::string &msg) -> int {
        seterr(err, msg);
        delete m;
        return code;
    };
    uint32_t magic = 0, hidden = 0;
    if (!rd(&magic, 4) || magic != 0x756d7867u) // model.cpp:101-106
        return fail(UMX_HOST_ERR_FORMAT, "invalid model data (bad magic)");
    if (!rd(&hidden, 4) || hidden == 0 || hidden > 65536 || hidden % 2)
        return fail(UMX_HOST_ERR_FORMAT, "invalid hidden size");
    m->hidden = (int)hidden;
    m->progress = 0.3f; // model.cpp:184-188
    int target = 0;
    while (pos < b.size()) // model.cpp:201-232
    {
        Rec r;
        int32_t name_len = 0;
        if (!rd(&r.scale, 4) || !rd(&r.offset, 4) || !rd(&r.n_dims, 4) || !rd(&name_len, 4))
            return fail(UMX_HOST_ERR_FORMAT, "truncated tensor header");
        if (r.n_dims < 1 || r.n_dims > 2 || name_len < 1 || name_len > 128)
            return fail(UMX_HOST_ERR_FORMAT, "corrupt tensor header");
        r.ne[0] = r.ne[1] = 1;
        size_t nel = 1;
        for (int i = 0; i < r.n_dims; ++i)
        {
            int32_t d;
            if (!rd(&d, 4) || d < 1)
                return fail(UMX_HOST_ERR_FORMAT, "corrupt tensor dims");
            r.ne[i] = d;
            nel *= (size_t)d;
        }
        r.name.resize(name_len);
        if (!rd(&r.name[0], name_len))
            return fail(UMX_HOST_ERR_FORMAT, "truncated tensor name");
        if (target >= 4)
            return fail(UMX_HOST_ERR_FORMAT, "more than 4 targets in model file");
        int e0, e1;
        if (!expected_shape(r.name, m->hidden, &e0, &e1)) // model.cpp:541-546
            return fail(UMX_HOST_ERR_FORMAT, "failed to load " + r.name + " (unknown tensor)");
        if (r.ne[0] != e0 || r.ne[1] != e1) // model.cpp:582-591
        {
            char buf[200];
            snprintf(buf, sizeof buf, "tensor '%s' has wrong size in model file: [%d, %d], expected [%d, %d]",
                     r.name.c_str(), r.ne[0], r.ne[1], e0, e1);
            return fail(UMX_HOST_ERR_FORMAT, buf);
        }
        r.dtype = is_u16(r.name) ? UMX_DTYPE_U16 : UMX_DTYPE_U8;
        r.nbytes = nel * (r.dtype == UMX_DTYPE_U16 ? 2 : 1);
        if (pos + r.nbytes > b.size())
            return fail(UMX_HOST_ERR_FORMAT, "truncated tensor data for " + r.name);
        r.data_off = pos;
        pos += r.nbytes;
        r.target = target;
        m->data_bytes += r.nbytes;
        m->recs.push_back(r);
        m->progress += 0.004f;        // model.cpp:551
        if (r.name == "bn3.running_var") // model.cpp:530-539
            ++target;
    }
    // the reference never checks completeness (a short file just leaves matrices unset); the
    // engine needs all 4 x 43, so say so here rather than at first use
    std::map<std::string, int> seen[4];
    for (const Rec &r : m->recs)
        seen[r.target][r.name]++;
    for (int t = 0; t < 4; ++t)
        if (seen[t].size() != 43)
            return fail(UMX_HOST_ERR_FORMAT, "model file does not hold 43 distinct tensors for target " +
                                                 std::to_string(t));
    m->views.resize(m->recs.size());
    for (size_t i = 0; i < m->recs.size(); ++i)
    {
        const Rec &r = m->recs[i];
        umx_tensor_view &v = m->views[i];
        v.name = r.name.c_str();
        v.target = r.target;
        v.dtype = r.dtype;
        v.n_dims = r.n_dims;
        v.ne[0] = r.ne[0];
        v.ne[1] = r.ne[1];
        v.scale = r.scale;
        v.offset = r.offset;
        v.data = m->blob.data() + r.data_off;
    }
    m->progress = 1.0f; // model.cpp:555
    *out = m;
    return UMX_OK;
}

extern "C" void umx_model_free(umx_model *m) { delete m; }
extern "C" int umx_model_hidden(const umx_model *m) { return m ? m->hidden : 0; }
extern "C" int umx_model_n_tensors(const umx_model *m) { return m ? (int)m->views.size() : 0; }
extern "C" const umx_tensor_view *umx_model_views(const umx_model *m) { return m ? m->views.data() : nullptr; }
extern "C" size_t umx_model_data_bytes(const umx_model *m) { return m ? m->data_bytes : 0; }
extern "C" float umx_model_load_progress(const umx_model *m) { return m ? m->progress : 0.f; }

extern "C" long umx_model_dequantize(const umx_model *m, int target, const char *name, float *dst, size_t cap)
{
    if (!m || !name)
        return -1;
    for (const Rec &r : m->recs)
        if (r.target == target && r.name == name)
        {
            const size_t nel = (size_t)r.ne[0] * r.ne[1];
            if (!dst)
                return (long)nel;
            if (cap < nel)
                return -3;
            const unsigned char *q = m->blob.data() + r.data_off;
            if (r.dtype == UMX_DTYPE_U16)
                for (size_t i = 0; i < nel; ++i)
                {
                    uint16_t v;
                    memcpy(&v, q + 2 * i, 2);
                    dst[i] = (float)v * r.scale + r.offset; // model.cpp:656-662
                }
            else
                for (size_t i = 0; i < nel; ++i)
                    dst[i] = (float)q[i] * r.scale + r.offset; // model.cpp:610-616
            return (long)nel;
        }
    return -2;
}
