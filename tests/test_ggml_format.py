"""Weight-file format (scripts/convert-umx-pth-to-ggml.py + model.cpp:93-665): quantiser, writer,
three independent readers (Python, oracle C++, host C++), and the loader's error behaviour."""
import gzip
import struct

import numpy as np
import pytest


def test_quantiser_matches_reference_formula(pkg):
    a = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    for qt, qmax in ((np.uint8, 255), (np.uint16, 65535)):
        q, scale, offset = pkg.ggml.quantize(a, qt)
        assert q.dtype == qt and offset == a.min()
        assert np.isclose(scale, (a.max() - a.min()) / (qmax - 1))
        assert q.max() <= qmax - 1 and q.min() == 0  # min->0, max->qmax-1 (convert script :19)
        back = pkg.ggml.dequantize(q, scale, offset)
        assert np.abs(back - a).max() <= scale * 0.5 + 4e-7 * np.abs(a).max()  # half a step + fp32 rounding


def test_tensor_table(pkg, po):
    names = pkg.ggml.tensor_names()
    assert len(names) == 43 and names[-1] == "bn3.running_var" and names == po.tensor_names()
    assert pkg.ggml.tensor_shape("fc1.weight", 1024) == (1024, 2974)
    assert pkg.ggml.tensor_shape("lstm.weight_hh_l2_reverse", 1024) == (2048, 512)
    assert pkg.ggml.tensor_shape("fc3.weight", 1024) == (4098, 1024)
    total = sum(int(np.prod(pkg.ggml.tensor_shape(n, 1024))) for n in names) * 4
    assert total == 113_077_920  # UMX-L parameter count


def test_u16_names(pkg):
    assert pkg.ggml.is_u16("fc2.weight") and pkg.ggml.is_u16("bn3.running_var")
    assert not pkg.ggml.is_u16("fc1.weight") and not pkg.ggml.is_u16("lstm.bias_hh_l0")


def test_three_readers_agree_bitwise(pkg, po, model_small):
    path, om, targets = model_small
    hm = pkg.HostModel(path)
    assert hm.hidden == 128 and hm.n_tensors == 172
    for t in range(4):
        for i, nm in enumerate(pkg.ggml.tensor_names()):
            a, b, c = hm.dequantize(t, nm), om.tensor(t, i), targets[t][nm]["f32"].ravel()
            assert (a == b).all() and (a == c).all(), nm


def test_views_keep_file_dims_reversed(pkg, model_small):
    path, _, _ = model_small
    hm = pkg.HostModel(path)
    views, n = hm.views()
    byname = {(views[i].target, views[i].name.decode()): views[i] for i in range(n)}
    v = byname[(3, "fc1.weight")]
    assert (v.n_dims, v.ne[0], v.ne[1], v.dtype) == (2, 2974, 128, pkg.DTYPE_U8)
    v = byname[(0, "fc3.weight")]
    assert (v.ne[0], v.ne[1], v.dtype) == (128, 4098, pkg.DTYPE_U16)
    v = byname[(1, "bn1.running_var")]
    assert (v.n_dims, v.ne[0], v.dtype) == (1, 128, pkg.DTYPE_U8)


def _raw(path):
    return gzip.open(path, "rb").read()


@pytest.mark.parametrize("breakage", ["magic", "truncated", "unknown", "shape", "missing_target"])
def test_loader_error_behaviour(pkg, po, model_small, tmp_path, breakage):
    """model.cpp returns false on bad magic (:101-106), unknown name (:541-546) and wrong shape
    (:582-591); both restated loaders must refuse the same files."""
    path, _, _ = model_small
    raw = bytearray(_raw(path))
    if breakage == "magic":
        raw[0] ^= 0xFF
    elif breakage == "truncated":
        raw = raw[:len(raw) // 2]
    elif breakage == "unknown":
        i = raw.index(b"input_mean")
        raw[i:i + 10] = b"input_meaX"
    elif breakage == "shape":
        i = raw.index(b"input_mean") - 4  # the single ne[] entry precedes the name
        raw[i:i + 4] = struct.pack("<i", 1486)
    elif breakage == "missing_target":
        i = raw.rindex(b"input_mean") - 20  # start of the last target's first record
        raw = raw[:i]
    p = tmp_path / "bad.bin"
    p.write_bytes(bytes(raw))
    with pytest.raises(pkg.HostError):
        pkg.HostModel(p)
    with pytest.raises(RuntimeError):
        po.Model.load(p)


def test_plain_and_gzip_files_load_identically(pkg, model_small, tmp_path):
    path, _, _ = model_small
    p = tmp_path / "plain.bin"
    p.write_bytes(_raw(path))
    a, b = pkg.HostModel(path), pkg.HostModel(p)
    assert (a.dequantize(2, "fc2.weight") == b.dequantize(2, "fc2.weight")).all()


# ---- the reference's OWN quantiser (convert-umx-pth-to-ggml.py:13-34), run in the build container by
# tests/golden/make_golden.py quant: the first reference-generated fixture behind the STFT
def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def quant_ref():
    from pathlib import Path
    return np.load(Path(__file__).parent / "golden" / "quantizer_ref.npz")


def test_quantiser_reproduces_the_reference_quantiser_bit_for_bit(pkg, quant_ref):
    g = quant_ref
    W = pkg.ggml.synth_weights(int(g["hidden"]), seed=int(g["wseed"]))[0]
    names = [str(n) for n in g["names"]]
    assert names == pkg.ggml.tensor_names()
    for i, nm in enumerate(names):
        q, scale, offset = pkg.ggml.quantize(W[nm], np.uint16 if pkg.ggml.is_u16(nm) else np.uint8)
        assert q.dtype == (np.uint16 if pkg.ggml.is_u16(nm) else np.uint8), nm  # the converter's rule, :146-150
        assert np.float32(scale).tobytes() == g["scale"][i].tobytes() and np.float32(offset).tobytes() == g["offset"][i].tobytes(), nm
        assert _sha(q) == str(g["q_sha256"][i]), nm
        deq = pkg.ggml.dequantize(q, scale, offset)
        assert deq.dtype == np.float32 and _sha(deq) == str(g["deq_sha256"][i]), nm
        if W[nm].ndim == 1:  # small tensors are stored in full
            assert (q == g["q/" + nm]).all() and (deq == g["deq/" + nm]).all(), nm


def test_quantiser_special_cases_match_the_reference_quantiser(pkg, quant_ref):
    import importlib.util
    import warnings
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("make_golden", Path(__file__).parent / "golden" / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # constant tensors divide 0 by 0 in the reference, too
        for nm, a in mg.quant_special_cases().items():
            for qt in (np.uint8, np.uint16):
                key = f"special/{nm}/{np.dtype(qt).name}"
                q, scale, offset = pkg.ggml.quantize(a, qt)
                assert (q == quant_ref[key + "/q"]).all(), key
                assert np.float32(scale).tobytes() == quant_ref[key + "/scale"].tobytes(), key  # bits: NaN-safe
                assert np.float32(offset).tobytes() == quant_ref[key + "/offset"].tobytes(), key
                assert pkg.ggml.dequantize(q, scale, offset).tobytes() == quant_ref[key + "/deq"].tobytes(), key


def test_loaders_dequantise_a_reference_quantised_file_bit_for_bit(pkg, po, quant_ref, tmp_path):
    """A model file whose first target holds the fixture's tensors: host/model.cpp's loader and the oracle's loader
    (model.cpp:578-665 restated twice) return exactly what the reference's `dequantize` returned for them."""
    g = quant_ref
    H = int(g["hidden"])
    path = str(tmp_path / "quantref.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=int(g["wseed"])), H)
    hm, om = pkg.HostModel(path), po.Model.load(path)
    _, targets = pkg.ggml.read_model(path)
    for i, nm in enumerate(pkg.ggml.tensor_names()):
        assert _sha(targets[0][nm]["q"]) == str(g["q_sha256"][i]), nm  # the bytes in the file are the reference's q
        assert _sha(hm.dequantize(0, nm)) == str(g["deq_sha256"][i]), nm
        assert _sha(np.asarray(om.tensor(0, i), np.float32)) == str(g["deq_sha256"][i]), nm
