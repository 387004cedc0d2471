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
