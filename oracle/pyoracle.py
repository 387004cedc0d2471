"""ctypes view of oracle/liboracle.so -- the CPU restatement of sevagh/umx.cpp's hot path.

TEST INFRASTRUCTURE ONLY (see oracle/umx_oracle.h): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product package.
Layouts returned to Python are numpy-friendly:
  spec / y      complex64 (2, T, 2049)      (reference: Eigen ColMajor (2,T,2049))
  mix_mag, mags float32   (2, T, 2049)
  waveforms     float32   (2, n)            (reference: MatrixXf(2, n) == interleaved stereo)
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
NB, CROP, NFFT, HOP = 2049, 1487, 4096, 1024
TENSORS_PER_TARGET = 43

_fp = C.POINTER(C.c_float)


def build(force=False):
    """make -C oracle (g++ only; seconds).  Libraries that exist are rebuilt when stale only in the build container
    (where /root/reference exists); on the GPU box the prebuilt files that travelled with the snapshot are used as
    they are -- importing the checker never compiles anything there unless a library is missing altogether."""
    missing = not (HERE / "liboracle.so").exists() or not (HERE / "liboracle_fast.so").exists()
    stale = not missing and Path("/root/reference").exists() and \
        max((HERE / f).stat().st_mtime for f in ("umx_oracle.cpp", "umx_oracle.h")) > (HERE / "liboracle.so").stat().st_mtime
    if force or missing or stale:
        subprocess.check_call(["make", "-C", str(HERE)], stdout=subprocess.DEVNULL)


class Taps(C.Structure):
    _fields_ = [("spec", _fp), ("mix_mag", _fp), ("x", _fp), ("fc1_out", _fp * 4),
                ("lstm_out", _fp * 4), ("mask", _fp * 4), ("target_mag", _fp * 4), ("y", _fp * 4), ("fc2_out", _fp * 4)]


_fast_path = [None]  # set by build_fast_native(): the timed CPU baseline built for THIS machine's cores


def build_fast_native():
    """The reference's Release flag set (CMakeLists.txt:17-23: -O3 -march=native -ffast-math) compiled ON THE MACHINE THAT
    TIMES IT, into oracle/_build/ (ignored by git and rebuilt per box): the prebuilt liboracle_fast.so travels from the
    build container, whose -march=native is another CPU's.  Returns a description of what will be loaded; falls back to
    the shipped library when there is no compiler.  Must run before the first lib(fast=True)."""
    out = HERE / "_build" / "liboracle_fast_native.so"
    try:
        out.parent.mkdir(exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                               "-ffast-math", "-DNDEBUG", "-o", str(out), str(HERE / "umx_oracle.cpp"), "-lz"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        _fast_path[0] = out
        _libs.pop(True, None)
        return "built on this machine (-O3 -march=native -ffast-math -fopenmp)"
    except Exception as e:  # noqa: BLE001 - the baseline is a report
        return f"prebuilt in the build container (native build failed: {type(e).__name__})"


def _load(fast=False):
    build()
    lib = C.CDLL(str((_fast_path[0] or HERE / "liboracle_fast.so") if fast else HERE / "liboracle.so"))
    lib.oracle_tensor_name.restype = C.c_char_p
    lib.oracle_tensor_name.argtypes = [C.c_int]
    lib.oracle_tensor_numel.restype = C.c_size_t
    lib.oracle_tensor_numel.argtypes = [C.c_int, C.c_int]
    lib.oracle_model_load.restype = C.c_void_p
    lib.oracle_model_load.argtypes = [C.c_char_p, C.c_char_p]
    lib.oracle_model_from_arrays.restype = C.c_void_p
    lib.oracle_model_from_arrays.argtypes = [C.c_int, C.POINTER(_fp)]
    lib.oracle_model_free.argtypes = [C.c_void_p]
    lib.oracle_model_hidden.argtypes = [C.c_void_p]
    lib.oracle_model_tensor.restype = _fp
    lib.oracle_model_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.oracle_hann_window.argtypes = [_fp]
    lib.oracle_window_sumsq.argtypes = [C.c_int, _fp]
    lib.oracle_nb_frames.argtypes = [C.c_int]
    lib.oracle_stft.argtypes = [_fp, C.c_int, C.c_int, _fp]
    lib.oracle_istft.argtypes = [_fp, C.c_int, C.c_int, _fp]
    lib.oracle_rfft4096.argtypes = [_fp, _fp]
    lib.oracle_irfft4096.argtypes = [_fp, _fp]
    lib.oracle_stream_state_floats.restype = C.c_size_t
    lib.oracle_stream_state_floats.argtypes = [C.c_int]
    lib.oracle_lstm_forward.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int, _fp, _fp]
    lib.oracle_target_network.argtypes = [C.c_void_p, C.c_int, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp, _fp]
    lib.oracle_target_network_ex.argtypes = [C.c_void_p, C.c_int, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]
    lib.oracle_wiener.argtypes = [_fp, C.POINTER(_fp), C.c_int, C.POINTER(_fp)]
    lib.oracle_umx_inference.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _fp, C.POINTER(_fp),
                                         C.c_int, C.POINTER(Taps)]
    lib.oracle_split_inference.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(_fp), C.c_int]
    lib.oracle_shift_inference.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(_fp), C.c_int]
    lib.oracle_set_num_threads.argtypes = [C.c_int]
    return lib


_libs = {}


def lib(fast=False):
    if fast not in _libs:
        _libs[fast] = _load(fast)
    return _libs[fast]


def _p(a):
    return a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- layout helpers: reference ColMajor (2,T,B) <-> numpy C-order (2,T,B) -------------------
def cm_to_np(flat, T, complex_=False):
    """flat ColMajor (2,T,2049) buffer -> C-order array [c, t, b]."""
    if complex_:
        a = flat.view(np.complex64).reshape((2, T, NB), order="F")
    else:
        a = flat.reshape((2, T, NB), order="F")
    return np.ascontiguousarray(a)


def np_to_cm(a):
    """C-order [c,t,b] -> flat ColMajor float32 buffer (complex -> interleaved re,im)."""
    f = np.asfortranarray(a).ravel(order="F")
    if np.iscomplexobj(f):
        f = f.astype(np.complex64).view(np.float32)
    return np.ascontiguousarray(f, dtype=np.float32)


def interleave(w):
    """(2, n) -> interleaved stereo buffer (== Eigen ColMajor MatrixXf(2,n))."""
    return np.ascontiguousarray(np.asarray(w, dtype=np.float32).T).ravel()


def deinterleave(buf, n):
    return np.ascontiguousarray(buf.reshape(n, 2).T)


def tensor_names():
    return [lib().oracle_tensor_name(i).decode() for i in range(TENSORS_PER_TARGET)]


class Model:
    """Oracle-side model (dequantised fp32 tensors in PyTorch layout)."""

    def __init__(self, handle, fast=False):
        if not handle:
            raise RuntimeError("oracle model handle is NULL")
        self.h = handle
        self.fast = fast
        self.hidden = lib(fast).oracle_model_hidden(handle)

    @classmethod
    def load(cls, path, fast=False):
        err = C.create_string_buffer(256)
        h = lib(fast).oracle_model_load(str(path).encode(), err)
        if not h:
            raise RuntimeError("oracle_model_load: " + err.value.decode())
        return cls(h, fast)

    @classmethod
    def from_arrays(cls, hidden, tensors, fast=False):
        """tensors: list of 4 dicts name -> fp32 array (PyTorch shapes)."""
        names = tensor_names()
        keep = [_f32(tensors[t][nm]).ravel() for t in range(4) for nm in names]
        for k, a in enumerate(keep):
            want = lib(fast).oracle_tensor_numel(k % TENSORS_PER_TARGET, hidden)
            assert a.size == want, (names[k % TENSORS_PER_TARGET], a.size, want)
        arr = (_fp * len(keep))(*[_p(a) for a in keep])
        return cls(lib(fast).oracle_model_from_arrays(hidden, arr), fast)

    def tensor(self, target, idx):
        n = lib(self.fast).oracle_tensor_numel(idx, self.hidden)
        ptr = lib(self.fast).oracle_model_tensor(self.h, target, idx)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def __del__(self):
        try:
            lib(self.fast).oracle_model_free(self.h)
        except Exception:
            pass


def hann_window():
    w = np.empty(NFFT, np.float32)
    lib().oracle_hann_window(_p(w))
    return w


def nb_frames(n_buf):
    return lib().oracle_nb_frames(n_buf)


def window_sumsq(T):
    nw = np.empty(NFFT + HOP * (T - 1), np.float32)
    lib().oracle_window_sumsq(T, _p(nw))
    return nw


def rfft4096(x):
    x = _f32(x)
    out = np.empty(2 * NB, np.float32)
    lib().oracle_rfft4096(_p(x), _p(out))
    return out.view(np.complex64)


def irfft4096(X):
    X = np.ascontiguousarray(X, np.complex64).view(np.float32)
    out = np.empty(NFFT, np.float32)
    lib().oracle_irfft4096(_p(X), _p(out))
    return out


def stft(wave, n_buf=None):
    """wave (2,n) -> complex64 [2,T,2049]; T = n_buf/1024+1 (dsp.hpp:48)."""
    wave = np.asarray(wave, np.float32)
    n = wave.shape[1]
    n_buf = n if n_buf is None else n_buf
    T = nb_frames(n_buf)
    a = interleave(wave)
    spec = np.empty(2 * 2 * T * NB, np.float32)
    lib().oracle_stft(_p(a), n, n_buf, _p(spec))
    return cm_to_np(spec, T, True)


def istft(spec, n, n_buf=None):
    n_buf = n if n_buf is None else n_buf
    flat = np_to_cm(spec)
    out = np.empty(2 * n, np.float32)
    lib().oracle_istft(_p(flat), n, n_buf, _p(out))
    return deinterleave(out, n)


def stream_state(hidden):
    return np.zeros(lib().oracle_stream_state_floats(hidden), np.float32)


def lstm_forward(model, target, x, state):
    """x (T,H) -> (T,H); state: one target's [3,2,2,H/2] block, updated in place."""
    x = _f32(x)
    T = x.shape[0]
    out = np.empty_like(x)
    assert state.dtype == np.float32 and state.flags.c_contiguous
    lib(model.fast).oracle_lstm_forward(model.h, target, _p(x), T, _p(state), _p(out))
    return out


def target_network(model, target, x, mix_mag, state):
    """inference.cpp:70-186 for one target.  x (T,2974), mix_mag [2,T,B], state = that target's [3][2][2][H/2] block
    (updated in place) -> dict of fc1 / lstm / fc2 (T,H), mask (T,4098), target_mag [2,T,B]."""
    x = np.ascontiguousarray(x, np.float32)
    T, H = x.shape[0], model.hidden
    mm = np_to_cm(np.asarray(mix_mag, np.float32))
    o = {k: np.empty(n, np.float32) for k, n in (("fc1", T * H), ("lstm", T * H), ("fc2", T * H), ("mask", T * 2 * NB),
                                                ("target_mag", 2 * T * NB))}
    lib(model.fast).oracle_target_network_ex(model.h, target, _p(x), _p(mm), T, _p(state), _p(o["fc1"]), _p(o["lstm"]),
                                             _p(o["fc2"]), _p(o["mask"]), _p(o["target_mag"]))
    return {"fc1": o["fc1"].reshape(T, H), "lstm": o["lstm"].reshape(T, H), "fc2": o["fc2"].reshape(T, H),
            "mask": o["mask"].reshape(T, 2 * NB), "target_mag": cm_to_np(o["target_mag"], T)}


def wiener(spec, target_mags):
    """spec complex [2,T,B], target_mags 4 x [2,T,B] -> 4 x complex [2,T,B]."""
    T = spec.shape[1]
    s = np_to_cm(spec)
    mags = [np_to_cm(m) for m in target_mags]
    ys = [np.empty(2 * 2 * T * NB, np.float32) for _ in range(4)]
    lib().oracle_wiener(_p(s), (_fp * 4)(*[_p(m) for m in mags]), T, (_fp * 4)(*[_p(y) for y in ys]))
    return [cm_to_np(y, T, True) for y in ys]


def umx_inference(model, wave, n_buf=None, state=None, flags=0, want_taps=False):
    """One segment (inference.cpp:12-207). Returns (4 x (2,n) waveforms, taps dict or None)."""
    wave = np.asarray(wave, np.float32)
    n = wave.shape[1]
    n_buf = n if n_buf is None else n_buf
    T = nb_frames(n_buf)
    H = model.hidden
    if state is None:
        state = stream_state(H)
    a = interleave(wave)
    outs = [np.empty(2 * n, np.float32) for _ in range(4)]
    taps = None
    keep = {}
    if want_taps:
        taps = Taps()
        keep["spec"] = np.empty(4 * T * NB, np.float32)
        keep["mix_mag"] = np.empty(2 * T * NB, np.float32)
        keep["x"] = np.empty(T * 2 * CROP, np.float32)
        taps.spec, taps.mix_mag, taps.x = _p(keep["spec"]), _p(keep["mix_mag"]), _p(keep["x"])
        for name, sz in (("fc1_out", T * H), ("lstm_out", T * H), ("fc2_out", T * H), ("mask", T * 2 * NB),
                         ("target_mag", 2 * T * NB), ("y", 4 * T * NB)):
            keep[name] = [np.empty(sz, np.float32) for _ in range(4)]
            arr = getattr(taps, name)
            for t in range(4):
                arr[t] = _p(keep[name][t])
    lib(model.fast).oracle_umx_inference(model.h, _p(a), n, n_buf, _p(state),
                                         (_fp * 4)(*[_p(o) for o in outs]), flags,
                                         C.byref(taps) if taps is not None else None)
    res = [deinterleave(o, n) for o in outs]
    if not want_taps:
        return res, None
    d = {
        "spec": cm_to_np(keep["spec"], T, True),
        "mix_mag": cm_to_np(keep["mix_mag"], T),
        "x": keep["x"].reshape(T, 2 * CROP),
        "fc1_out": [k.reshape(T, H) for k in keep["fc1_out"]],
        "lstm_out": [k.reshape(T, H) for k in keep["lstm_out"]],
        "fc2_out": [k.reshape(T, H) for k in keep["fc2_out"]],
        "mask": [k.reshape(T, 2 * NB) for k in keep["mask"]],
        "target_mag": [cm_to_np(k, T) for k in keep["target_mag"]],
        "y": [cm_to_np(k, T, True) for k in keep["y"]],
    }
    return res, d


def split_inference(model, wave, segment_samples, flags=0):
    wave = np.asarray(wave, np.float32)
    L = wave.shape[1]
    a = interleave(wave)
    outs = [np.empty(2 * L, np.float32) for _ in range(4)]
    lib(model.fast).oracle_split_inference(model.h, _p(a), L, segment_samples,
                                           (_fp * 4)(*[_p(o) for o in outs]), flags)
    return [deinterleave(o, L) for o in outs]


def shift_inference(model, wave, segment_samples, offset=4033, flags=0):
    wave = np.asarray(wave, np.float32)
    L = wave.shape[1]
    a = interleave(wave)
    outs = [np.empty(2 * L, np.float32) for _ in range(4)]
    lib(model.fast).oracle_shift_inference(model.h, _p(a), L, segment_samples, offset,
                                           (_fp * 4)(*[_p(o) for o in outs]), flags)
    return [deinterleave(o, L) for o in outs]


def set_num_threads(n, fast=False):
    lib(fast).oracle_set_num_threads(n)
