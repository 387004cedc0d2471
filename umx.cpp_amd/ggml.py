"""The reference's ggml-style UMX weight file: writer, reader, quantiser, synthetic weights.

Restates scripts/convert-umx-pth-to-ggml.py (writer: :13-26 quantiser, :99 magic, :127 hidden,
:146-160 per-tensor record) and the reader side of src/model.cpp:93-232 so synthetic weights can
be produced in exactly the format `load_umx_model` consumes (the real UMX-L file is a git-LFS
pointer in the reference).

File: i32 magic 0x756d7867, i32 hidden, then for targets bass, drums, other, vocals 43 records
  {f32 scale, f32 offset, i32 n_dims, i32 name_len, i32 ne[n_dims] (PyTorch shape reversed),
   name bytes, quantised data in PyTorch row-major order}
u16 for names containing fc2/fc3/bn2/bn3, u8 otherwise (:146); EVERY tensor is quantised.
"""
import gzip
import struct

import numpy as np

MAGIC = 0x756D7867
NB, CROP = 2049, 1487
TARGETS = ["bass", "drums", "other", "vocals"]  # convert script :104


def tensor_names():
    """Order of the 43 tensors of one target (bn3.running_var must close a target: model.cpp:530-539)."""
    names = ["input_mean", "input_scale", "output_scale", "output_mean", "fc1.weight",
             "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var"]
    for layer in range(3):
        for suffix in ("", "_reverse"):
            for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                names.append(f"lstm.{w}_l{layer}{suffix}")
    names += ["fc2.weight", "bn2.weight", "bn2.bias", "bn2.running_mean", "bn2.running_var",
              "fc3.weight", "bn3.weight", "bn3.bias", "bn3.running_mean", "bn3.running_var"]
    assert len(names) == 43
    return names


def tensor_shape(name, hidden):
    """PyTorch shape of a tensor for a given hidden size (model.cpp:118-135)."""
    H, Hl = hidden, hidden // 2
    if name in ("input_mean", "input_scale"):
        return (CROP,)
    if name in ("output_mean", "output_scale"):
        return (NB,)
    if name == "fc1.weight":
        return (H, 2 * CROP)
    if name == "fc2.weight":
        return (H, 2 * H)
    if name == "fc3.weight":
        return (2 * NB, H)
    if name.startswith("bn1") or name.startswith("bn2"):
        return (H,)
    if name.startswith("bn3"):
        return (2 * NB,)
    if "weight_ih" in name:
        return (4 * Hl, H)
    if "weight_hh" in name:
        return (4 * Hl, Hl)
    if "bias_" in name:
        return (4 * Hl,)
    raise KeyError(name)


def is_u16(name):
    return any(x in name for x in ("bn2", "bn3", "fc2", "fc3"))  # convert script :146


def quantize(array, qtype=np.uint8):
    """convert script :13-26: scale=(max-min)/(qmax-1), offset=min, q=round((x-offset)/scale)."""
    array = np.asarray(array, np.float32)
    mn, mx = np.min(array), np.max(array)
    scale = (mx - mn) / float(np.iinfo(qtype).max - 1)
    offset = mn
    q = np.round((array - offset) / scale).astype(qtype)
    return q, np.float32(scale), np.float32(offset)


def dequantize(q, scale, offset):
    """model.cpp:610-616 / 656-662: q*scale+offset evaluated in fp32."""
    return (q.astype(np.float32) * np.float32(scale) + np.float32(offset)).astype(np.float32)


def synth_weights(hidden=1024, seed=0):
    """Seeded UMX-shaped weights with PyTorch-default-like ranges so activations stay
    out of saturation: Linear/LSTM ~U(+-1/sqrt(fan_in)), BN weight ~U(.5,1.5), running_var > 0."""
    rng = np.random.default_rng(seed)
    out = []
    for _t in range(4):
        d = {}
        for name in tensor_names():
            shp = tensor_shape(name, hidden)
            if name == "input_mean":
                a = rng.uniform(-0.2, 0.2, shp)
            elif name == "input_scale":
                a = rng.uniform(0.02, 0.08, shp)  # |STFT| of 0.5-peak audio is O(10..100)
            elif name == "output_scale":
                a = rng.uniform(0.5, 1.5, shp)
            elif name == "output_mean":
                a = rng.uniform(0.0, 1.0, shp)
            elif name.endswith("running_var"):
                a = rng.uniform(0.5, 1.5, shp)
            elif name.endswith("running_mean"):
                a = rng.normal(0.0, 0.1, shp)
            elif name.startswith("bn") and name.endswith("weight"):
                a = rng.uniform(0.5, 1.5, shp)
            elif name.startswith("bn") and name.endswith("bias"):
                a = rng.uniform(-0.1, 0.1, shp)
            elif name.startswith("lstm"):
                k = 1.0 / np.sqrt(hidden // 2)
                a = rng.uniform(-k, k, shp)
            else:  # fc weights, fan_in = shape[1]
                k = 1.0 / np.sqrt(shp[1])
                a = rng.uniform(-k, k, shp)
            d[name] = a.astype(np.float32)
        out.append(d)
    return out


def write_model(path, weights, hidden, compress=True):
    """Write 4 dicts name->fp32 array in the reference format; returns the dequantised weights
    (what every consumer of the file sees), same structure."""
    blob = bytearray()
    blob += struct.pack("i", MAGIC)
    blob += struct.pack("i", hidden)
    deq = []
    for t in range(4):
        dd = {}
        for name in tensor_names():
            data = np.asarray(weights[t][name], np.float32)
            assert data.shape == tensor_shape(name, hidden), (name, data.shape)
            q, scale, offset = quantize(data, np.uint16 if is_u16(name) else np.uint8)
            nm = name.encode("utf-8")
            blob += struct.pack("ffii", scale, offset, data.ndim, len(nm))
            for i in range(data.ndim):
                blob += struct.pack("i", data.shape[data.ndim - 1 - i])
            blob += nm
            blob += q.tobytes()
            dd[name] = dequantize(q, scale, offset)
        deq.append(dd)
    if compress:
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(bytes(blob))
    else:
        with open(path, "wb") as f:
            f.write(bytes(blob))
    return deq


def read_model(path):
    """Python reader (independent of the C++ loaders; used to cross-check them).
    Returns (hidden, [4 dicts name -> dict(q=quantised array, scale, offset, f32=dequantised)])."""
    with open(path, "rb") as f:
        head = f.read(2)
    opener = gzip.open if head == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        buf = f.read()
    pos = 0
    magic, hidden = struct.unpack_from("ii", buf, pos)
    pos += 8
    if magic != MAGIC:
        raise ValueError("bad magic")
    targets, cur = [], {}
    while pos < len(buf):
        scale, offset, n_dims, name_len = struct.unpack_from("ffii", buf, pos)
        pos += 16
        ne = struct.unpack_from("i" * n_dims, buf, pos)
        pos += 4 * n_dims
        name = buf[pos:pos + name_len].decode()
        pos += name_len
        n = int(np.prod(ne))
        dt = np.uint16 if is_u16(name) else np.uint8
        q = np.frombuffer(buf, dt, n, pos).reshape(tuple(reversed(ne)))
        pos += n * np.dtype(dt).itemsize
        cur[name] = dict(q=q, scale=np.float32(scale), offset=np.float32(offset),
                         f32=dequantize(q, scale, offset))
        if name == "bn3.running_var":
            targets.append(cur)
            cur = {}
    return hidden, targets


def synth_audio(n, seed=0):
    """Seeded synthetic 44.1 kHz stereo (2,n) fp32: ~16 log-spaced sinusoids with slow
    envelopes + 0.02 N(0,1), peak 0.5; right = left mixed with an independent copy at 0.3."""
    def one(rng):
        t = np.arange(n, dtype=np.float64) / 44100.0
        x = np.zeros(n)
        for f in np.geomspace(55.0, 12000.0, 16):
            ph = rng.uniform(0, 2 * np.pi)
            env = 0.5 * (1 + np.sin(2 * np.pi * rng.uniform(0.05, 0.5) * t + rng.uniform(0, 6.28)))
            x += env * np.sin(2 * np.pi * f * t + ph) / 16.0
        x += 0.02 * rng.standard_normal(n)
        return x
    a = one(np.random.default_rng(seed))
    b = one(np.random.default_rng(seed + 1000))
    left = a
    right = a + 0.3 * b
    peak = max(np.abs(left).max(), np.abs(right).max())
    return (np.stack([left, right]) * (0.5 / peak)).astype(np.float32)
