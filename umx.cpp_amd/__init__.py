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
FLAG_LSTM_FORCE_SAFE = 0x40
FLAG_LSTM_PROFILE = 0x80
FLAG_PRECISE_ACT = 0x1000
FLAG_DEBUG_LSTM_ABORT = 0x2000
FLAG_RESET_SEGMENTS = 0x4000  # whole-track calls: every segment from a zero LSTM state, segments as lanes of one call (declared deviation)


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
    try:  # if torch is going to be used in this process it must bring ITS HIP runtime in first: loading
        import torch  # the system libamdhip64 before torch's own copy makes torch report "No HIP GPUs"
        torch.cuda.is_available()
    except Exception:  # noqa: BLE001 - torch is optional plumbing
        pass
    path = Path(os.environ.get("UMX_HIP_LIB", HERE / "libumx_hip.so"))  # override = kernel-variant A/B runs
    if not path.exists():
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(str(path))
    lib.umx_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(TensorView), C.c_int]
    lib.umx_hip_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(TensorView), C.c_int,
                                      C.c_uint]
    lib.umx_hip_create_tracks.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(TensorView), C.c_int,
                                          C.c_uint, C.c_int]
    lib.umx_hip_n_tracks.argtypes = [C.c_void_p]
    lib.umx_hip_pipeline_depth.argtypes = [C.c_void_p]
    lib.umx_hip_lstm_is_batched.argtypes = [C.c_void_p]
    lib.umx_hip_debug_f16_bits.argtypes = [C.c_float]
    lib.umx_hip_debug_f16_bits.restype = C.c_uint
    lib.umx_hip_track_stream_reset.argtypes = [C.c_void_p, C.c_int]
    lib.umx_hip_track_stream_get.argtypes = [C.c_void_p, C.c_int, _fp]
    lib.umx_hip_track_stream_set.argtypes = [C.c_void_p, C.c_int, _fp]
    lib.umx_hip_infer_segment_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_uint]
    lib.umx_hip_infer_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_uint]
    lib.umx_hip_infer_batch_async.argtypes = lib.umx_hip_infer_batch.argtypes
    lib.umx_hip_infer_batch_device.argtypes = lib.umx_hip_infer_batch.argtypes
    lib.umx_hip_separate_tracks.argtypes = [C.c_void_p, C.c_int, C.POINTER(_fp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_fp),
                                            C.c_uint, C.c_void_p, C.c_void_p]
    lib.umx_hip_order_after.argtypes = [C.c_void_p, C.c_void_p]
    lib.umx_hip_order_before.argtypes = [C.c_void_p, C.c_void_p]
    lib.umx_hip_weight_bytes.restype = C.c_size_t
    lib.umx_hip_weight_bytes.argtypes = [C.c_void_p]
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
    lib.umx_hip_stage_times_slot.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), _fp, C.c_int]
    lib.umx_hip_stage_kernel_times_slot.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int]
    lib.umx_hip_lstm_was_persistent.argtypes = [C.c_void_p]
    lib.umx_hip_lstm_mode.argtypes = [C.c_void_p]
    lib.umx_hip_debug_lstm_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.umx_hip_debug_wiener_bins.argtypes = [C.c_int, _fp, _fp, _fp, C.c_float, _fp]
    lib.umx_hip_lstm_kernel_name.restype = C.c_char_p
    lib.umx_hip_lstm_kernel_name.argtypes = [C.c_void_p]
    lib.umx_hip_gemm_kernel_name.restype = C.c_char_p
    lib.umx_hip_gemm_kernel_name.argtypes = [C.c_void_p, C.c_int]
    lib.umx_hip_debug_lstm_placement.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    lib.umx_hip_stream_layer_floats.restype = C.c_size_t
    lib.umx_hip_stream_layer_floats.argtypes = [C.c_void_p]
    lib.umx_hip_stream_get_layer.argtypes = [C.c_void_p, C.c_int, _fp]
    lib.umx_hip_stream_set_layer.argtypes = [C.c_void_p, C.c_int, _fp]
    lib.umx_hip_segment_begin.argtypes = [C.c_void_p, _fp, C.c_int, C.c_uint]
    lib.umx_hip_segment_lstm_layer.argtypes = [C.c_void_p, C.c_int]
    lib.umx_hip_segment_end.argtypes = [C.c_void_p, C.POINTER(_fp)]
    lib.umx_hip_debug_lds_guard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint)]
    lib.umx_hip_split_inference.argtypes = [C.c_void_p, _fp, C.c_int, C.POINTER(_fp), C.c_uint, C.c_void_p, C.c_void_p]
    lib.umx_hip_shift_inference.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(_fp), C.c_uint, C.c_void_p,
                                            C.c_void_p]
    _hip = lib
    return lib


CREATE_QUANTISED_RESIDENT = 0x1
CREATE_GEMM_F32 = 0x4
CREATE_DEQUANTISE_AT_LOAD = 0x8
CREATE_LSTM_BATCHED = 0x10
CREATE_U8_DEQUANT = 0x20
CREATE_GEMM_STAGED = 0x40
CREATE_GEMM_PLANES = 0x80
MAX_TRACKS = 64

HIP_SYMBOLS = ["umx_hip_create", "umx_hip_create_ex", "umx_hip_create_tracks", "umx_hip_n_tracks", "umx_hip_lstm_is_batched",
               "umx_hip_track_stream_reset", "umx_hip_track_stream_get", "umx_hip_track_stream_set",
               "umx_hip_infer_segment_async", "umx_hip_infer_batch", "umx_hip_infer_batch_async",
               "umx_hip_infer_batch_device", "umx_hip_order_after", "umx_hip_order_before",
               "umx_hip_separate_tracks", "umx_hip_phase_stream", "umx_hip_stream_state_device", "umx_hip_segment_begin_device", "umx_hip_segment_end_device",
               "umx_hip_weight_stems_device", "umx_hip_track_accumulate_device", "umx_hip_track_normalise_device", "umx_hip_weight_bytes", "umx_hip_destroy", "umx_hip_last_error", "umx_hip_stream_floats",
               "umx_hip_stream_reset", "umx_hip_stream_get", "umx_hip_stream_set", "umx_hip_infer_segment",
               "umx_hip_infer_segment_device", "umx_hip_sync", "umx_hip_stream_handle", "umx_hip_nb_frames",
               "umx_hip_segment_samples", "umx_hip_hidden", "umx_hip_read_tap", "umx_hip_stage_times",
               "umx_hip_stage_times_slot", "umx_hip_stage_kernel_times_slot",
               "umx_hip_lstm_was_persistent", "umx_hip_lstm_mode", "umx_hip_lstm_kernel_name", "umx_hip_gemm_kernel_name", "umx_hip_debug_wiener_bins", "umx_hip_debug_lstm_profile", "umx_hip_debug_lstm_placement",
               "umx_hip_stream_layer_floats", "umx_hip_stream_get_layer", "umx_hip_stream_set_layer",
               "umx_hip_segment_begin", "umx_hip_segment_lstm_layer", "umx_hip_segment_end",
               "umx_hip_split_inference", "umx_hip_shift_inference", "umx_hip_debug_lds_guard", "umx_hip_debug_f16_bits",
               "umx_hip_segment_masks_device", "umx_hip_target_mag_device", "umx_hip_segment_finish_device", "umx_hip_gate_reserve", "umx_hip_segment_discard", "umx_hip_pipeline_depth"]


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

    def __init__(self, targets, hidden, segment_samples=SEGMENT_SAMPLES, device=0, quantised=True,
                 quantised_resident=True, gemm=None, tracks=1, lstm_batched=False, u8_dequant=False):
        """quantised: hand the file's u8/u16 bytes (+ scale/offset) to the engine instead of fp32 arrays;
        quantised_resident: keep them that way in HBM (the default; BASELINE config 5) or expand them at load;
        gemm: "planes" (fp16 matrix cores, operands pre-split into fp16 planes, LDS-DMA staging: the default with tracks > 1
        or lstm_batched) or "bf16x3" (bf16 terms, both operands split while every tile is staged: the default of the
        single-track engine); "f32" (the fp32-MFMA flavour of rounds 1-2) is refused;
        tracks: independent track lanes (1..64) run together per call (infer_batch*);
        lstm_batched: use the batched (matrix-core) LSTM kernel also on a 1-track context."""
        self.lib = hip_lib()
        views, self._keep = views_from_file_tensors(targets, quantised)
        h = C.c_void_p()
        rc = self.lib.umx_hip_create_tracks(C.byref(h), device, hidden, segment_samples, views, len(views),
                                            (0 if quantised_resident else CREATE_DEQUANTISE_AT_LOAD) |
                                            {"f32": CREATE_GEMM_F32, "bf16x3": CREATE_GEMM_STAGED, "planes": CREATE_GEMM_PLANES}.get(
                                                gemm or os.environ.get("UMX_GEMM", ""), 0) |
                                            (CREATE_LSTM_BATCHED if lstm_batched else 0) |
                                            (CREATE_U8_DEQUANT if (u8_dequant or os.environ.get("UMX_U8") == "dequant") else 0), tracks)
        if rc != UMX_OK:
            raise UmxError(rc, self.lib.umx_hip_last_error(None).decode())
        self.h = h
        self.hidden = hidden
        self.N = segment_samples
        self.T = self.lib.umx_hip_nb_frames(h)
        self.tracks = tracks

    @classmethod
    def from_file(cls, path, segment_samples=SEGMENT_SAMPLES, device=0, quantised_resident=True, gemm=None, tracks=1,
                  lstm_batched=False, quantised=True, u8_dequant=False):
        hidden, targets = ggml.read_model(path)
        return cls(targets, hidden, segment_samples, device, quantised=quantised, quantised_resident=quantised_resident,
                   gemm=gemm, tracks=tracks, lstm_batched=lstm_batched, u8_dequant=u8_dequant)

    def lstm_is_batched(self):
        return bool(self.lib.umx_hip_lstm_is_batched(self.h))

    def pipeline_depth(self):
        """Segments in flight together: that many consecutive asynchronous calls need distinct buffers."""
        return int(self.lib.umx_hip_pipeline_depth(self.h))

    def weight_bytes(self):
        return int(self.lib.umx_hip_weight_bytes(self.h))

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

    def track_stream_reset(self, track=-1):
        self._check(self.lib.umx_hip_track_stream_reset(self.h, track))

    def track_stream_get(self, track):
        a = np.empty(self.lib.umx_hip_stream_floats(self.h), np.float32)
        self._check(self.lib.umx_hip_track_stream_get(self.h, track, a.ctypes.data_as(_fp)))
        return a

    def track_stream_set(self, track, a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.size == self.lib.umx_hip_stream_floats(self.h)
        self._check(self.lib.umx_hip_track_stream_set(self.h, track, a.ctypes.data_as(_fp)))

    def stream_get_layer(self, layer):
        """(h, c) of LSTM layer `layer`, all chains: [4 targets][2 dirs][2: h, c][hidden/2]."""
        a = np.empty(self.lib.umx_hip_stream_layer_floats(self.h), np.float32)
        self._check(self.lib.umx_hip_stream_get_layer(self.h, layer, a.ctypes.data_as(_fp)))
        return a

    def stream_set_layer(self, layer, a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.size == self.lib.umx_hip_stream_layer_floats(self.h)
        self._check(self.lib.umx_hip_stream_set_layer(self.h, layer, a.ctypes.data_as(_fp)))

    # --- one segment phase by phase (exact multi-GPU carry, multigpu.separate_track_carry_mode) ---
    def segment_begin(self, wave, flags=0):
        wave = np.asarray(wave, np.float32)
        self._phase_n = wave.shape[1]
        a = np.ascontiguousarray(wave.T).ravel()
        self._check(self.lib.umx_hip_segment_begin(self.h, a.ctypes.data_as(_fp), self._phase_n, flags))

    def segment_lstm_layer(self, layer):
        self._check(self.lib.umx_hip_segment_lstm_layer(self.h, layer))

    def segment_end(self):
        n = self._phase_n
        outs = [np.empty(2 * n, np.float32) for _ in range(4)]
        arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs])
        self._check(self.lib.umx_hip_segment_end(self.h, arr))
        return [np.ascontiguousarray(o.reshape(n, 2).T) for o in outs]

    # --- whole track on the device (umx.cpp:99-295) ---
    def separate(self, wave, flags=0, shift_offset=None):
        """(2,L) host array -> 4 x (2,L): split_inference with the track resident in HBM; shift_offset: None =
        no shift buffer, n >= 0 = shift_inference with that offset."""
        wave = np.asarray(wave, np.float32)
        L = wave.shape[1]
        a = np.ascontiguousarray(wave.T).ravel()
        outs = [np.empty(2 * L, np.float32) for _ in range(4)]
        self.separate_interleaved(a, L, outs, flags, shift_offset)
        return [np.ascontiguousarray(o.reshape(L, 2).T) for o in outs]

    def separate_many(self, waves, flags=0, shift_offsets=None):
        """Several tracks at once, one per track lane (umx_hip_separate_tracks): list of (2,L_i) -> list of [4 x (2,L_i)]."""
        nt = len(waves)
        ins = [np.ascontiguousarray(np.asarray(w, np.float32).T).ravel() for w in waves]
        Ls = [np.asarray(w).shape[1] for w in waves]
        outs = [[np.empty(2 * L, np.float32) for _ in range(4)] for L in Ls]
        a = (_fp * nt)(*[x.ctypes.data_as(_fp) for x in ins])
        o = (_fp * (4 * nt))(*[x.ctypes.data_as(_fp) for t4 in outs for x in t4])
        sh = (C.c_int * nt)(*[(-1 if s is None else s) for s in (shift_offsets or [None] * nt)])
        self._check(self.lib.umx_hip_separate_tracks(self.h, nt, a, (C.c_int * nt)(*Ls), sh, o, flags, None, None))
        return [[np.ascontiguousarray(x.reshape(L, 2).T) for x in t4] for t4, L in zip(outs, Ls)]

    def separate_interleaved(self, a, L, outs, flags=0, shift_offset=None):
        """The bare C call: a = (2,L) interleaved float32, outs = 4 preallocated float32[2L]; returns seconds."""
        import time
        arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs])
        t0 = time.perf_counter()
        if shift_offset is None:
            rc = self.lib.umx_hip_split_inference(self.h, a.ctypes.data_as(_fp), L, arr, flags, None, None)
        else:
            rc = self.lib.umx_hip_shift_inference(self.h, a.ctypes.data_as(_fp), L, shift_offset, arr, flags, None, None)
        dt = time.perf_counter() - t0
        self._check(rc)
        return dt

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

    def infer_batch(self, waves, flags=0):
        """waves: list (one per track lane) of (2,n_i) arrays or None (lane idle) -> list of [4 x (2,n_i)] or None."""
        nb = len(waves)
        aud, ns, outs = (C.c_void_p * nb)(), (C.c_int * nb)(), (C.c_void_p * (4 * nb))()
        keep = []
        for i, wv in enumerate(waves):
            if wv is None:
                aud[i], ns[i] = None, 0
                continue
            wv = np.asarray(wv, np.float32)
            a = np.ascontiguousarray(wv.T).ravel()
            o = [np.empty(2 * wv.shape[1], np.float32) for _ in range(4)]
            keep.append((a, o))
            aud[i], ns[i] = a.ctypes.data, wv.shape[1]
            for t in range(4):
                outs[4 * i + t] = o[t].ctypes.data
        self._check(self.lib.umx_hip_infer_batch(self.h, nb, aud, ns, outs, flags))
        res, k = [], 0
        for wv in waves:
            if wv is None:
                res.append(None)
                continue
            n = keep[k][0].size // 2
            res.append([np.ascontiguousarray(o.reshape(n, 2).T) for o in keep[k][1]])
            k += 1
        return res

    def infer_batch_ptrs(self, audio_ptrs, ns, out_ptrs, flags=0, where="device"):
        """Raw pointers: audio_ptrs[i] (0 = lane idle), ns[i], out_ptrs[4*i + t]; where = "device" (HBM buffers) or
        "host_async" (host buffers, pinned for overlap: H2D + kernels + D2H queued).  Asynchronous: call sync()."""
        nb = len(audio_ptrs)
        aud = (C.c_void_p * nb)(*[p or None for p in audio_ptrs])
        n_ = (C.c_int * nb)(*ns)
        outs = (C.c_void_p * (4 * nb))(*[p or None for p in out_ptrs])
        fn = self.lib.umx_hip_infer_batch_device if where == "device" else self.lib.umx_hip_infer_batch_async
        self._check(fn(self.h, nb, aud, n_, outs, flags))

    def infer_segment_device(self, audio_ptr, n, out_ptrs, flags=0):
        """Raw device pointers (e.g. torch tensor .data_ptr()); asynchronous, call sync()."""
        arr = (C.c_void_p * 4)(*out_ptrs)
        self._check(self.lib.umx_hip_infer_segment_device(self.h, C.c_void_p(audio_ptr), n, arr, flags))

    def sync(self):
        self._check(self.lib.umx_hip_sync(self.h))

    def last_error(self):
        return self.lib.umx_hip_last_error(self.h).decode()

    def lstm_was_persistent(self):
        return bool(self.lib.umx_hip_lstm_was_persistent(self.h))

    def lstm_mode(self):
        """0 stepwise, 1 persistent (placement-independent hand-off), 2 persistent (intra-XCD hand-off)."""
        return self.lib.umx_hip_lstm_mode(self.h)

    def lstm_profile(self):
        buf = (C.c_ulonglong * 48)()
        self._check(self.lib.umx_hip_debug_lstm_profile(self.h, buf))
        a = np.array(buf[:], dtype=np.uint64).reshape(3, 2, 8)
        return a

    def gemm_kernel_name(self, mode):
        """The GEMM kernel the last call launched for stage `mode` (0 fc1, 1 W_ih, 2 fc2, 3 fc3)."""
        return self.lib.umx_hip_gemm_kernel_name(self.h, mode).decode()

    def lstm_kernel_name(self):
        """The recurrence kernel of the last LSTM layer launch (umx_hip_lstm_kernel_name)."""
        return self.lib.umx_hip_lstm_kernel_name(self.h).decode()

    def lstm_placement(self, n=256):
        """(xcc, chain, slice, HW_ID) of every workgroup of the last profiled one-track recurrence launch."""
        buf = (C.c_ulonglong * n)()
        self._check(self.lib.umx_hip_debug_lstm_placement(self.h, buf, n))
        a = np.array(buf[:], dtype=np.uint64)
        return [(int(v >> 48) & 15, int(v >> 40) & 255, int(v >> 32) & 255, int(v) & 0xffffffff) for v in a]

    def tap(self, what, target=0):
        n = self.lib.umx_hip_read_tap(self.h, what.encode(), target, None, 0)
        if n < 0:
            raise UmxError(ERR_ARG, f"tap {what!r} unavailable ({n})")
        buf = np.empty(n, np.float32)
        got = self.lib.umx_hip_read_tap(self.h, what.encode(), target, buf.ctypes.data_as(_fp), n)
        if got != n:
            raise UmxError(ERR_HIP, f"tap {what!r} failed ({got})")
        T, H = self.T, self.hidden
        what = what.split("@")[0].split("#")[0]
        if what in ("lstm_l0", "lstm_l1"):
            return buf.reshape(T, H)
        if what == "proj":
            return buf.reshape(T, 4 * H)
        if what in ("spec", "y"):
            return buf.view(np.complex64).reshape(2, T, NB)
        if what in ("mix_mag", "target_mag"):
            return buf.reshape(2, T, NB)
        if what == "x":
            return buf.reshape(T, KX)
        if what in ("fc1", "lstm", "fc2"):
            return buf.reshape(T, H)
        if what == "mask":
            return buf.reshape(T, NOUT)
        return buf

    def stage_times(self, slot=None):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        if slot is None:
            n = self.lib.umx_hip_stage_times(self.h, names, ms, 32)
        else:
            n = self.lib.umx_hip_stage_times_slot(self.h, slot, names, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def stage_kernel_times(self, slot=None):
        """Like stage_times, but a GEMM stage of a plane context without its split kernel (umx_hip_stage_kernel_times_slot)."""
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = self.lib.umx_hip_stage_times_slot(self.h, 0, names, (C.c_float * 32)(), 32)
        m = self.lib.umx_hip_stage_kernel_times_slot(self.h, -1 if slot is None else slot, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(min(n, m))}


# ------------------------------------------------------------------ C++17 host library (umx_host.h)
HOST_SYMBOLS = ["umx_model_load", "umx_model_free", "umx_model_hidden", "umx_model_n_tensors", "umx_model_views",
                "umx_model_data_bytes", "umx_model_load_progress", "umx_model_dequantize", "umx_wav_load",
                "umx_wav_free", "umx_wav_write_f32", "umx_split_inference", "umx_shift_inference",
                "umx_segment_plan", "umx_transition_weight", "umx_split_inference_carry", "umx_split_inference_targets"]

SEGMENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, C.c_int, C.POINTER(_fp))
RESET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_float, C.c_void_p)


PH_BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, C.c_int)
PH_LAYER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
PH_END_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_fp))
PH_STATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _fp)
P2P_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, C.c_size_t, C.c_int)


class PhasedBackend(C.Structure):
    """include/umx_host.h: umx_phased_backend"""
    _fields_ = [("begin", PH_BEGIN_FN), ("layer", PH_LAYER_FN), ("end", PH_END_FN), ("get_layer", PH_STATE_FN),
                ("set_layer", PH_STATE_FN), ("layer_floats", C.c_size_t), ("user", C.c_void_p)]


TG_BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _fp, C.c_int, C.c_uint)
TG_STATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, _fp)
TG_VOID_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
TG_MAG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _fp)


class TargetBackend(C.Structure):
    """include/umx_host.h: umx_target_backend"""
    _fields_ = [("begin", TG_BEGIN_FN), ("layer", PH_LAYER_FN), ("get_state", TG_STATE_FN), ("set_state", TG_STATE_FN),
                ("masks", TG_VOID_FN), ("get_mag", TG_MAG_FN), ("set_mag", TG_MAG_FN), ("finish", PH_END_FN), ("discard", TG_VOID_FN),
                ("target_layer_floats", C.c_size_t), ("mag_floats", C.c_size_t), ("user", C.c_void_p)]


class P2P(C.Structure):
    """include/umx_host.h: umx_p2p"""
    _fields_ = [("send", P2P_FN), ("recv", P2P_FN), ("user", C.c_void_p)]


class Backend(C.Structure):
    """include/umx_host.h: umx_backend"""
    _fields_ = [("segment", SEGMENT_FN), ("reset", RESET_FN), ("user", C.c_void_p)]


_host = None


def host_lib():
    global _host
    if _host is not None:
        return _host
    path = HERE / "libumx_host.so"
    if not path.exists():
        raise ImportError(f"{path} is missing: run __graft_entry__.build()")
    lib = C.CDLL(str(path))
    lib.umx_model_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p]
    lib.umx_model_free.argtypes = [C.c_void_p]
    lib.umx_model_hidden.argtypes = [C.c_void_p]
    lib.umx_model_n_tensors.argtypes = [C.c_void_p]
    lib.umx_model_views.restype = C.POINTER(TensorView)
    lib.umx_model_views.argtypes = [C.c_void_p]
    lib.umx_model_data_bytes.restype = C.c_size_t
    lib.umx_model_data_bytes.argtypes = [C.c_void_p]
    lib.umx_model_load_progress.restype = C.c_float
    lib.umx_model_load_progress.argtypes = [C.c_void_p]
    lib.umx_model_dequantize.restype = C.c_long
    lib.umx_model_dequantize.argtypes = [C.c_void_p, C.c_int, C.c_char_p, _fp, C.c_size_t]
    lib.umx_wav_load.argtypes = [C.c_char_p, C.POINTER(_fp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p]
    lib.umx_wav_free.argtypes = [_fp]
    lib.umx_wav_write_f32.argtypes = [C.c_char_p, _fp, C.c_int, C.c_char_p]
    lib.umx_split_inference.argtypes = [C.POINTER(Backend), _fp, C.c_int, C.c_int, C.POINTER(_fp), PROGRESS_FN,
                                        C.c_void_p, C.c_char_p]
    lib.umx_shift_inference.argtypes = [C.POINTER(Backend), _fp, C.c_int, C.c_int, C.c_int, C.POINTER(_fp),
                                        PROGRESS_FN, C.c_void_p, C.c_char_p]
    lib.umx_split_inference_carry.argtypes = [C.POINTER(PhasedBackend), C.POINTER(P2P), C.c_int, C.c_int, _fp, C.c_int, C.c_int,
                                              C.POINTER(_fp), C.c_char_p]
    lib.umx_split_inference_targets.argtypes = [C.POINTER(TargetBackend), C.POINTER(P2P), C.c_int, C.c_int, _fp, C.c_int, C.c_int,
                                                C.POINTER(_fp), C.c_char_p]
    lib.umx_segment_plan.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.umx_transition_weight.restype = C.c_float
    lib.umx_transition_weight.argtypes = [C.c_int, C.c_int, C.c_int]
    _host = lib
    return lib


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"umx_host error {code}: {msg}")
        self.code = code


class HostModel:
    """load_umx_model (model.cpp:42-574) through the C++ host loader; tensors stay quantised."""

    def __init__(self, path):
        self.lib = host_lib()
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = self.lib.umx_model_load(str(path).encode(), C.byref(h), err)
        if rc:
            raise HostError(rc, err.value.decode())
        self.h = h
        self.hidden = self.lib.umx_model_hidden(h)
        self.n_tensors = self.lib.umx_model_n_tensors(h)

    def views(self):
        return self.lib.umx_model_views(self.h), self.n_tensors

    def dequantize(self, target, name):
        n = self.lib.umx_model_dequantize(self.h, target, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        a = np.empty(n, np.float32)
        self.lib.umx_model_dequantize(self.h, target, name.encode(), a.ctypes.data_as(_fp), n)
        return a

    def data_bytes(self):
        return self.lib.umx_model_data_bytes(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.umx_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def engine_from_host_model(hm, segment_samples=SEGMENT_SAMPLES, device=0):
    """The CLI's path: C++ loader views -> umx_hip_create (no Python copy of the weights)."""
    eng = Engine.__new__(Engine)
    eng.lib = hip_lib()
    views, n = hm.views()
    h = C.c_void_p()
    rc = eng.lib.umx_hip_create(C.byref(h), device, hm.hidden, segment_samples, views, n)
    if rc != UMX_OK:
        raise UmxError(rc, eng.lib.umx_hip_last_error(None).decode())
    eng.h, eng.hidden, eng.N, eng._keep = h, hm.hidden, segment_samples, None
    eng.T = eng.lib.umx_hip_nb_frames(h)
    return eng


def wav_load(path):
    """load_audio (dsp.cpp:18-77): -> ((2,n) float32, channels in file)."""
    lib = host_lib()
    p, n, ch = _fp(), C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    rc = lib.umx_wav_load(str(path).encode(), C.byref(p), C.byref(n), C.byref(ch), err)
    if rc:
        raise HostError(rc, err.value.decode())
    a = np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()
    lib.umx_wav_free(p)
    return np.ascontiguousarray(a.T), ch.value


def wav_write(path, wave):
    """write_audio_file (dsp.cpp:80-101): (2,n) -> stereo float32 WAV."""
    a = np.ascontiguousarray(np.asarray(wave, np.float32).T)
    err = C.create_string_buffer(256)
    rc = host_lib().umx_wav_write_f32(str(path).encode(), a.ctypes.data_as(_fp), a.shape[0], err)
    if rc:
        raise HostError(rc, err.value.decode())


def segment_plan(length, segment_samples=SEGMENT_SAMPLES):
    lib = host_lib()
    n = lib.umx_segment_plan(length, segment_samples, None, None, 0)
    offs, lens = (C.c_int * n)(), (C.c_int * n)()
    lib.umx_segment_plan(length, segment_samples, offs, lens, n)
    return list(offs), list(lens)


def make_backend(segment_fn, reset_fn=None):
    """Wrap Python callables as a umx_backend: segment_fn((2,n) array) -> 4 x (2,n) arrays."""
    def _seg(_user, audio, n, out):
        try:
            wave = np.ctypeslib.as_array(audio, shape=(n, 2)).T
            res = segment_fn(np.ascontiguousarray(wave))
            for t in range(4):
                dst = np.ctypeslib.as_array(out[t], shape=(n, 2))
                dst[:, :] = np.asarray(res[t], np.float32).T
            return 0
        except Exception as e:  # noqa: BLE001 - surfaced as a backend error code
            print("segment backend raised:", repr(e))
            return 13

    def _reset(_user):
        if reset_fn is not None:
            reset_fn()
        return 0
    cb_seg, cb_reset = SEGMENT_FN(_seg), RESET_FN(_reset)
    be = Backend(cb_seg, cb_reset, None)
    be._keep = (cb_seg, cb_reset)
    return be


def _run_driver(kind, backend, wave, segment_samples, offset=None, progress=None):
    lib = host_lib()
    wave = np.asarray(wave, np.float32)
    L = wave.shape[1]
    a = np.ascontiguousarray(wave.T).ravel()
    outs = [np.empty(2 * L, np.float32) for _ in range(4)]
    arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs])
    err = C.create_string_buffer(256)
    pcb = PROGRESS_FN((lambda p, _u: progress(p)) if progress else (lambda p, _u: None))
    if kind == "split":
        rc = lib.umx_split_inference(C.byref(backend), a.ctypes.data_as(_fp), L, segment_samples, arr, pcb, None, err)
    else:
        rc = lib.umx_shift_inference(C.byref(backend), a.ctypes.data_as(_fp), L, segment_samples,
                                     -1 if offset is None else offset, arr, pcb, None, err)
    if rc:
        raise HostError(rc, err.value.decode())
    return [np.ascontiguousarray(o.reshape(L, 2).T) for o in outs]


def split_inference(backend, wave, segment_samples=SEGMENT_SAMPLES, progress=None):
    """umx.cpp:152-295 through the C++ host driver."""
    return _run_driver("split", backend, wave, segment_samples, progress=progress)


def shift_inference(backend, wave, segment_samples=SEGMENT_SAMPLES, offset=None, progress=None):
    """umx.cpp:99-150 through the C++ host driver (offset None = the reference's rand()%22050)."""
    return _run_driver("shift", backend, wave, segment_samples, offset=offset, progress=progress)


def engine_backend(eng, flags=0):
    return make_backend(lambda w: eng.infer_segment(w, flags), eng.stream_reset)


# ------------------------------------------------------------------ multi-GPU track driver (umx_mgpu.h)
MGPU_SYMBOLS = ["umx_mgpu_unique_id", "umx_mgpu_create", "umx_mgpu_create_ex", "umx_mgpu_destroy", "umx_mgpu_separate_track", "umx_mgpu_stats"]
MGPU_ID_BYTES = 640
MGPU_BY_TARGET, MGPU_LOOPBACK = 0x1, 0x2
_mgpu = None


def mgpu_lib():
    """libumx_mgpu.so: C++17 host over RCCL (loads librccl; a process that never shards a track does not need it)."""
    global _mgpu
    if _mgpu is not None:
        return _mgpu
    hip_lib()
    path = HERE / "libumx_mgpu.so"
    if not path.exists():
        raise ImportError(f"{path} is missing: run __graft_entry__.build()")
    lib = C.CDLL(str(path))
    lib.umx_mgpu_unique_id.argtypes = [C.c_char_p, C.c_char_p]
    lib.umx_mgpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    lib.umx_mgpu_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_uint, C.c_char_p]
    lib.umx_mgpu_destroy.argtypes = [C.c_void_p]
    lib.umx_mgpu_separate_track.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.POINTER(_fp), C.c_uint, C.c_char_p]
    lib.umx_mgpu_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    _mgpu = lib
    return lib


def mgpu_unique_id():
    buf = C.create_string_buffer(MGPU_ID_BYTES)
    err = C.create_string_buffer(256)
    rc = mgpu_lib().umx_mgpu_unique_id(buf, err)
    if rc:
        raise UmxError(rc, err.value.decode())
    return buf.raw


class MultiGpuTrack:
    """One track over `world` GPUs, exact (include/umx_mgpu.h): world = G target groups x P segment-pipeline stages; LSTM
    layer states, target magnitudes and weighted stems travel over RCCL point to point on device pointers.  ids: the
    bytes of mgpu_unique_id() made on rank 0 and handed to every rank (None when world == 1).  by_target: shard by source
    model as well (G = gcd(world, 4)); loopback (world 1 only): every transfer through a grouped RCCL self send + receive."""

    def __init__(self, engine, rank=0, world=1, ids=None, by_target=False, loopback=False):
        self.lib, self.eng, self.rank = mgpu_lib(), engine, rank
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = self.lib.umx_mgpu_create_ex(C.byref(h), engine.h, rank, world, ids, (MGPU_BY_TARGET if by_target else 0) |
                                         (MGPU_LOOPBACK if loopback else 0), err)
        if rc:
            raise UmxError(rc, err.value.decode())
        self.h = h

    def separate(self, wave, shift_offset=None, flags=0):
        """(2,L) host array on every rank -> 4 x (2,L) on rank 0 (None elsewhere); collective over the ranks."""
        wave = np.asarray(wave, np.float32)
        L = wave.shape[1]
        a = np.ascontiguousarray(wave.T).ravel()
        outs = [np.empty(2 * L, np.float32) for _ in range(4)] if self.rank == 0 else None
        arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs]) if outs else None
        err = C.create_string_buffer(256)
        rc = self.lib.umx_mgpu_separate_track(self.h, a.ctypes.data_as(_fp), L, -1 if shift_offset is None else shift_offset,
                                              arr, flags, err)
        if rc:
            raise UmxError(rc, err.value.decode())
        return [np.ascontiguousarray(o.reshape(L, 2).T) for o in outs] if outs else None

    def separate_interleaved(self, a, L, outs, shift_offset=None, flags=0):
        """The bare C call: a = (2,L) interleaved float32 on every rank, outs = 4 preallocated float32[2L] on rank 0 (None
        elsewhere); returns seconds."""
        import time
        arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs]) if outs else None
        err = C.create_string_buffer(256)
        t0 = time.perf_counter()
        rc = self.lib.umx_mgpu_separate_track(self.h, a.ctypes.data_as(_fp), L, -1 if shift_offset is None else shift_offset, arr, flags, err)
        dt = time.perf_counter() - t0
        if rc:
            raise UmxError(rc, err.value.decode())
        return dt

    def stats(self):
        """{rccl_ops, state_hops, magnitude_transfers, stem_transfers, retries} of the last track."""
        buf = (C.c_longlong * 5)()
        self.lib.umx_mgpu_stats(self.h, buf)
        return dict(zip(("rccl_ops", "state_hops", "magnitude_transfers", "stem_transfers", "retries"), [int(x) for x in buf]))

    def close(self):
        if getattr(self, "h", None):
            self.lib.umx_mgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
