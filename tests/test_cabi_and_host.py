"""C-ABI surface (every symbol the headers declare is exported; no compute without a GPU), and the
C++ host drivers (split / shift) against the oracle's restatement of umx.cpp:99-295."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared(header):
    src = (ROOT / "include" / header).read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(umx_[a-z0-9_]+)\s*\(", src)) - {"umx_segment_fn", "umx_reset_fn"})


def test_hip_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(str(ROOT / "umx.cpp_amd" / "libumx_hip.so"))
    names = _declared("umx_hip.h")
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(pkg.HIP_SYMBOLS) == names


def test_cu_reservations_are_requests_per_device_not_one_global_value(pkg):
    """umx_hip_gate_reserve (include/umx_hip.h): two multi-GPU drivers on one device each file a request; destroying
    one gives back ITS request only; giving back a request that was never filed is an error.  Pure host state: runs without a GPU."""
    lib = ctypes.CDLL(str(ROOT / "umx.cpp_amd" / "libumx_hip.so"))
    lib.umx_hip_gate_reserve.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.umx_hip_gate_reserve.restype = ctypes.c_int
    dev = 5  # any slot of the per-device table
    assert lib.umx_hip_gate_reserve(dev, 0) == 0        # clean slate
    assert lib.umx_hip_gate_reserve(dev, 16) == 0       # driver A
    assert lib.umx_hip_gate_reserve(dev, 8) == 0        # driver B
    assert lib.umx_hip_gate_reserve(dev, -16) == 0      # A is destroyed: B's request stays
    assert lib.umx_hip_gate_reserve(dev, -16) != 0      # nothing of that size left to give back
    assert lib.umx_hip_gate_reserve(dev, -8) == 0       # B is destroyed
    assert lib.umx_hip_gate_reserve(dev, -8) != 0
    assert lib.umx_hip_gate_reserve(-1, 4) != 0         # no such device


def test_host_side_fp16_encoding_of_weight_planes_is_round_to_nearest_even(pkg):
    """csrc/gemm_planes.h re-encodes weights as fp16 planes on the host at load time (u8 / u16 integers exactly, fp32
    weights as two split terms): its fp32 -> fp16 conversion against numpy's, over random values of every magnitude
    class (normal, subnormal, underflow, overflow, ties) and every integer a weight plane can hold."""
    lib = pkg.hip_lib()
    rng = np.random.default_rng(5)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
        np.arange(-32768, 32768, 7, dtype=np.float32), np.arange(-128, 128, dtype=np.float32) * 256.0,
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25, 2.0 ** -14,
                  2.0 ** -14 - 2.0 ** -25, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 2049.0, 2051.0, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([lib.umx_hip_debug_f16_bits(float(v)) for v in vals], dtype=np.uint16)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(float(vals[i]), hex(int(got[i])), hex(int(want[i]))) for i in bad[:5]]


def test_host_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(str(ROOT / "umx.cpp_amd" / "libumx_host.so"))
    names = _declared("umx_host.h")
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(pkg.HOST_SYMBOLS) == names


def test_product_has_no_cpu_fallback(pkg, model_small):
    """Without a GPU the engine must refuse loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    path, _, targets = model_small
    with pytest.raises(pkg.UmxError) as e:
        pkg.Engine(targets, 128, 16384)
    assert e.value.code == pkg.ERR_NODEVICE
    # nothing under the product package may name the checker: no import, dlopen, link or path of oracle/
    for p in (ROOT / "umx.cpp_amd").rglob("*"):
        if p.suffix in (".py", ".cpp", ".hip", ".h") or p.name == "Makefile":
            txt = p.read_text().lower()
            for needle in ("pyoracle", "liboracle", "oracle/", "import oracle", "from oracle"):
                assert needle not in txt, (p, needle)
    import sys
    assert not any(m.startswith("oracle") for m in sys.modules if "umx_cpp_amd" in str(getattr(sys.modules[m], "__file__", "")))


def test_segment_plan_and_weights(pkg):
    N = 2_646_000
    offs, lens = pkg.segment_plan(26_460_000 + 22050 - 4033, N)  # 600 s track after the shift pad
    assert len(offs) == 14 and offs[1] == 1_984_500 == int(0.75 * N)  # umx.cpp:181, SURVEY 8d
    assert lens[0] == N and lens[-1] < N
    lib = pkg.host_lib()
    assert lib.umx_transition_weight(0, N, N) == np.float32(1) / np.float32(N // 2)
    assert lib.umx_transition_weight(N // 2 - 1, N, N) == 1.0 == lib.umx_transition_weight(N // 2, N, N)
    assert lib.umx_transition_weight(N - 1, N, N) == np.float32(1) / np.float32(N // 2)
    # short last chunk: weight[:chunk_len] is the rising edge only (umx.cpp:246)
    assert lib.umx_transition_weight(999, 1000, N) == np.float32(1000) / np.float32(N // 2)


@pytest.fixture(scope="module")
def oracle_backend(pkg, po, model_small):
    _, om, _ = model_small
    N = 4 * 8192
    state = [po.stream_state(128)]

    def seg(w):
        return po.umx_inference(om, w, n_buf=N, state=state[0])[0]

    def reset():
        state[0] = po.stream_state(128)
    return pkg.make_backend(seg, reset), om, N


@pytest.mark.parametrize("length_factor", [0.4, 1.0, 2.3])
def test_split_inference_matches_oracle(pkg, po, oracle_backend, length_factor):
    """Short tracks (< one segment: the reference's UB case F4, here with sum_weight zeroed), exactly
    one segment, and several segments with a ragged tail.  Bit-exact: same ops, same order."""
    be, om, N = oracle_backend
    wave = pkg.ggml.synth_audio(int(N * length_factor), 9)
    a = pkg.split_inference(be, wave, N)
    b = po.split_inference(om, wave, N)
    for t in range(4):
        assert a[t].shape == wave.shape and (a[t] == b[t]).all()


def test_shift_inference_matches_oracle_and_default_offset(pkg, po, oracle_backend):
    be, om, N = oracle_backend
    wave = pkg.ggml.synth_audio(int(N * 1.3), 4)
    a = pkg.shift_inference(be, wave, N, offset=4033)  # unseeded glibc rand() % 22050 (umx.cpp:115)
    b = po.shift_inference(om, wave, N, 4033)
    assert all((a[t] == b[t]).all() for t in range(4))
    prog = []
    c = pkg.shift_inference(be, wave, N, offset=0, progress=prog.append)
    assert len(prog) == len(pkg.segment_plan(wave.shape[1] + 22050, N)[0]) and abs(prog[-1] - 1.0) < 1e-5
    assert np.abs(c[0] - a[0]).max() > 0  # a different shift is a different (valid) output


def test_backend_error_is_propagated(pkg):
    def boom(w):
        raise RuntimeError("nope")
    be = pkg.make_backend(boom)
    with pytest.raises(pkg.HostError) as e:
        pkg.split_inference(be, np.zeros((2, 5000), np.float32), 4096)
    assert e.value.code == 13


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """The gfx950 assembly of the engine, compiled once with the Makefile's flags (hipcc cross-compiles without a GPU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    mk = (ROOT / "umx.cpp_amd" / "Makefile").read_text()
    assert "-fno-slp-vectorize" in mk
    out = tmp_path_factory.mktemp("asm") / "engine.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
           f"-I{ROOT / 'include'}", "-S", "--cuda-device-only", "-o", str(out), str(ROOT / "umx.cpp_amd" / "csrc" / "engine.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def test_device_code_has_no_low_half_op_sel_on_packed_fp32(device_asm):
    """MI355X erratum found in round 1 (tools/pk_mfma_probe.hip): v_pk_add/mul/fma_f32 whose LOW result
    half selects the HIGH half of a source (op_sel bit set) return wrong values while a co-resident wave issues
    v_mfma_f32_32x32x16_bf16 -- which this engine's GEMMs do all the time.  The build therefore uses
    -fno-slp-vectorize and scalar horizontal adds; this test disassembles the device code and fails if a compiler
    or source change reintroduces such a form."""
    import re
    bad = []
    for line in device_asm.splitlines():
        m = re.match(r"\s*(v_pk_(?:add|mul|fma)_f32)\b.*\bop_sel:\[([01,]+)\]", line)
        if m and "1" in m.group(2):
            bad.append(line.strip())
    assert not bad, bad[:5]


def test_hot_kernels_keep_their_register_budgets(device_asm):
    """The kernels a 32-lane step spends its time in are written against a register budget: a source or
    compiler change that makes one of them spill -- the first persistent form of the fused Wiener kernel spilled 117 registers
    and nobody would have seen it in a parity test -- must fail here.  Read from the code object's metadata."""
    import re
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", device_asm):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))

    def find(*needles):
        hits = [(k, v) for k, v in meta.items() if all(n in k for n in needles)]
        assert len(hits) == 1, (needles, [k for k, _ in hits])
        return hits[0][1]

    # ping-pong plane GEMMs: 8 waves per CU = at most 256 registers; fc1 / W_ih / fc2 main loop + epilogue without a spill
    # (fc3's epilogue, cold code at the end of a tile, is allowed the handful it has had since round 3)
    for mode, nbp in ((0, 1), (1, 1), (2, 2)):
        vg, sp = find(f"gemm_planes_pp_kernelILi{mode}ELi{nbp}E")
        assert vg <= 256 and sp == 0, (mode, nbp, vg, sp)
    vg, sp = find("gemm_planes_pp_kernelILi3ELi2E")
    assert vg <= 256 and sp <= 16, (vg, sp)
    # persistent plane GEMMs (round 6): the main loop without a spill in any form; the epilogue trip of fc3 may keep its handful
    for mode, nbp in ((0, 1), (1, 1), (2, 2), (3, 2)):
        vg, sp = find(f"gemm_planes_ps_kernelILi{mode}ELi{nbp}E")
        assert vg <= 256 and sp <= 16, (mode, nbp, vg, sp)
    # octets of 8 lanes x column shards of 64 units (round 5; NO = octets per workgroup in turn): 128 registers of W_hh fragments per
    # wave; the PROLOGUE (256 byte loads per lane into those fragments) spills, the step loops (intra-XCD and sc1 protocol) must not
    for no in (1, 2):
        vg, sp = find(f"lstm_batch8_kernelILi512ELb0ELi{no}E")
        assert vg <= 256 and sp <= 128, (no, vg, sp)
        body = device_asm[device_asm.index(f"_ZN3umx18lstm_batch8_kernelILi512ELb0ELi{no}EEEvNS_9LstmBArgsE:"):]
        body = body[:body.index(".Lfunc_end")]
        steps = body.split("=>This Loop Header: Depth=1")[1:]  # ("Inner Loop Header" = the others)
        assert len(steps) == 2, (no, len(steps))
        for seg in steps:
            assert "v_mfma_f32_16x16x32_f16" in seg
            loop = seg[:seg.rindex("buffer_store_dwordx4")]  # through the last granule publication of the loop body
            assert "scratch_" not in loop, f"the step loop of lstm_batch8_kernel<{no}> touches scratch"
    # fused Wiener / inverse STFT / overlap-add: 1024 threads = at most 128
    vg, sp = find("wiener_istft_kernelILb1EE")
    assert vg <= 128 and sp == 0, (vg, sp)
    # single-track recurrence: two grids per CU in the cross-segment pipeline (gemm_common.h: 104 + 136 budget)
    hits = [v for k, v in meta.items() if "lstm_persistent_kernel" in k]
    assert hits and all(sp == 0 for _, sp in hits), hits
