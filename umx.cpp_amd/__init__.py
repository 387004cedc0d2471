"""umx.cpp_amd -- MI355X-native Open-Unmix (UMX-L) segment inference, drop-in for sevagh/umx.cpp's
src/inference.cpp hot path.

Python is plumbing only: this module is a ctypes view of the C-ABI in include/umx_hip.h (HIP
kernels, gfx950) and include/umx_host.h (C++17 host: ggml loader, wav I/O, segment drivers).
There is NO CPU fallback: if libumx_hip.so is missing or no GPU is present, construction fails
loudly.  The directory name has a dot, so import it with `__graft_entry__.load_package()`.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from . import ggml  # noqa: F401  (weight-file format + synthetic inputs)

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
NB, CROP, KX, NOUT, NFFT, HOP = 2049, 1487, 2976, 4098, 4096, 1024
SEGMENT_SAMPLES = 60 * 44100  # inference.hpp:13 x dsp.hpp:16

UMX_OK, ERR_ARG, ERR_HIP, ERR_MODEL, ERR_TIMEOUT, ERR_NODEVICE = 0, 1, 2, 3, 4, 5
DTYPE_F32, DTYPE_U8, DTYPE_U16 = 0, 1, 2
FLAG_NO_WIENER = 0x1
FLAG_LSTM_STEPWISE = 0x10
FLAG_DEBUG_TAPS = 0x20


def FLAG_SKIP_TARGET(t):
    return 0x100 << t


_fp = C.POINTER(C.c_float)


class TensorView(C.Structure):
    """include/umx_hip.h: umx_tensor_view"""
    _fields_ = [("name", C.c_char_p), ("target", C.c_int), ("dtype", C.c_int), ("n_dims", C.c_int),
                ("ne", C.c_int * 2), ("scale", C.c_float), ("offset", C.c_float), ("data", C.c_void_p)]


class UmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"umx_hip error {code}: {msg}")
        self.code = code


def build(verbose=False):
    """Compile every native library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", str(HERE), "all"], stdout=out)


_hip = None


def hip_lib():
    """Load libumx_hip.so (fails loudly: the product has no other compute path)."""
    global _hip
    if _hip is not None:
        return _hip
    path = HERE / "libumx_hip.so"
    if not path.exists():
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(str(path))
    lib.umx_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(TensorView), C.c_int]
    lib.umx_hip_destroy.argtypes = [C.c_void_p]
    lib.umx_hip_last_error.restype = C.c_char_p
    lib.umx_hip_last_error.argtypes = [C.c_void_p]
    lib.umx_hip_stream_floats.restype = C.c_size_t
    lib.umx_hip_stream_floats.argtypes = [C.c_void_p]
    lib.umx_hip_stream_reset.argtypes = [C.c_void_p]
    lib.umx_hip_stream_get.argtypes = [C.c_void_p, _fp]
    lib.umx_hip_stream_set.argtypes = [C.c_void_p, _fp]
    lib.umx_hip_infer_segment.argtypes = [C.c_void_p, _fp, C.c_int, C.POINTER(_fp), C.c_uint]
    lib.umx_hip_infer_segment_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_uint]
    lib.umx_hip_sync.argtypes = [C.c_void_p]
    lib.umx_hip_stream_handle.restype = C.c_void_p
    lib.umx_hip_stream_handle.argtypes = [C.c_void_p]
    lib.umx_hip_nb_frames.argtypes = [C.c_void_p]
    lib.umx_hip_segment_samples.argtypes = [C.c_void_p]
    lib.umx_hip_hidden.argtypes = [C.c_void_p]
    lib.umx_hip_read_tap.restype = C.c_long
    lib.umx_hip_read_tap.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _fp, C.c_size_t]
    lib.umx_hip_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), _fp, C.c_int]
    lib.umx_hip_lstm_was_persistent.argtypes = [C.c_void_p]
    _hip = lib
    return lib


HIP_SYMBOLS = ["umx_hip_create", "umx_hip_destroy", "umx_hip_last_error", "umx_hip_stream_floats",
               "umx_hip_stream_reset", "umx_hip_stream_get", "umx_hip_stream_set", "umx_hip_infer_segment",
               "umx_hip_infer_segment_device", "umx_hip_sync", "umx_hip_stream_handle", "umx_hip_nb_frames",
               "umx_hip_segment_samples", "umx_hip_hidden", "umx_hip_read_tap", "umx_hip_stage_times",
               "umx_hip_lstm_was_persistent"]


def views_from_file_tensors(targets, quantised=True):
    """Build umx_tensor_view[] from ggml.read_model() output.  quantised=True hands the u8/u16 bytes
    + scale/offset to the engine (it dequantises like model.cpp:610-616); False hands fp32."""
    keep, views = [], []
    for t, d in enumerate(targets):
        for name in ggml.tensor_names():
            rec = d[name]
            if isinstance(rec, dict) and quantised:
                arr = np.ascontiguousarray(rec["q"])
                dt = DTYPE_U16 if arr.dtype == np.uint16 else DTYPE_U8
                scale, offset = float(rec["scale"]), float(rec["offset"])
            else:
                arr = np.ascontiguousarray(rec["f32"] if isinstance(rec, dict) else rec, dtype=np.float32)
                dt, scale, offset = DTYPE_F32, 1.0, 0.0
            shp = arr.shape
            v = TensorView()
            nm = name.encode()
            keep.append((arr, nm))
            v.name, v.target, v.dtype, v.n_dims = nm, t, dt, len(shp)
            v.ne[0] = shp[-1]
            v.ne[1] = shp[0] if len(shp) == 2 else 1
            v.scale, v.offset, v.data = scale, offset, arr.ctypes.data
            views.append(v)
    return (TensorView * len(views))(*views), keep


class Engine:
    """One device context = the reference's (umx_model on device, stft_buffers, 4 x lstm_data).

    Mirrors the call shape of umx.cpp:160-227: create once per track, `infer_segment` per chunk,
    the streaming LSTM state carries over until `stream_reset`."""

    def __init__(self, targets, hidden, segment_samples=SEGMENT_SAMPLES, device=0, quantised=True):
        self.lib = hip_lib()
        views, self._keep = views_from_file_tensors(targets, quantised)
        h = C.c_void_p()
        rc = self.lib.umx_hip_create(C.byref(h), device, hidden, segment_samples, views, len(views))
        if rc != UMX_OK:
            raise UmxError(rc, self.lib.umx_hip_last_error(None).decode())
        self.h = h
        self.hidden = hidden
        self.N = segment_samples
        self.T = self.lib.umx_hip_nb_frames(h)

    @classmethod
    def from_file(cls, path, segment_samples=SEGMENT_SAMPLES, device=0):
        hidden, targets = ggml.read_model(path)
        return cls(targets, hidden, segment_samples, device)

    def _check(self, rc):
        if rc != UMX_OK:
            raise UmxError(rc, self.lib.umx_hip_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.umx_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- stream state (lstm.hpp:10-16) ---
    def stream_reset(self):
        self._check(self.lib.umx_hip_stream_reset(self.h))

    def stream_get(self):
        a = np.empty(self.lib.umx_hip_stream_floats(self.h), np.float32)
        self._check(self.lib.umx_hip_stream_get(self.h, a.ctypes.data_as(_fp)))
        return a

    def stream_set(self, a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.size == self.lib.umx_hip_stream_floats(self.h)
        self._check(self.lib.umx_hip_stream_set(self.h, a.ctypes.data_as(_fp)))

    # --- umx_inference (inference.cpp:12-207) ---
    def infer_segment(self, wave, flags=0):
        """wave (2,n) fp32 host array -> list of 4 (2,n) host arrays (H2D + kernels + D2H)."""
        wave = np.asarray(wave, np.float32)
        n = wave.shape[1]
        a = np.ascontiguousarray(wave.T).ravel()
        outs = [np.empty(2 * n, np.float32) for _ in range(4)]
        arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs])
        self._check(self.lib.umx_hip_infer_segment(self.h, a.ctypes.data_as(_fp), n, arr, flags))
        return [np.ascontiguousarray(o.reshape(n, 2).T) for o in outs]

    def infer_segment_device(self, audio_ptr, n, out_ptrs, flags=0):
        """Raw device pointers (e.g. torch tensor .data_ptr()); asynchronous, call sync()."""
        arr = (C.c_void_p * 4)(*out_ptrs)
        self._check(self.lib.umx_hip_infer_segment_device(self.h, C.c_void_p(audio_ptr), n, arr, flags))

    def sync(self):
        self._check(self.lib.umx_hip_sync(self.h))

    def lstm_was_persistent(self):
        return bool(self.lib.umx_hip_lstm_was_persistent(self.h))

    def tap(self, what, target=0):
        n = self.lib.umx_hip_read_tap(self.h, what.encode(), target, None, 0)
        if n < 0:
            raise UmxError(ERR_ARG, f"tap {what!r} unavailable ({n})")
        buf = np.empty(n, np.float32)
        got = self.lib.umx_hip_read_tap(self.h, what.encode(), target, buf.ctypes.data_as(_fp), n)
        if got != n:
            raise UmxError(ERR_HIP, f"tap {what!r} failed ({got})")
        T, H = self.T, self.hidden
        if what in ("spec", "y"):
            return buf.view(np.complex64).reshape(2, T, NB)
        if what in ("mix_mag", "target_mag"):
            return buf.reshape(2, T, NB)
        if what == "x":
            return buf.reshape(T, KX)
        if what in ("fc1", "lstm"):
            return buf.reshape(T, H)
        if what == "mask":
            return buf.reshape(T, NOUT)
        return buf

    def stage_times(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = self.lib.umx_hip_stage_times(self.h, names, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(n)}
