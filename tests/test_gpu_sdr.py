"""SURVEY 8(f)2 as a test: the real-weight validation tool (tests/sdr_check.py) run end to end on synthetic UMX-L-shaped
weights + the reference's shipped test track (test/data/gspi_stereo.wav, kept under tests/golden/), so that it is known to
work the day ggml-model-umxl-u8.bin.gz (sha256 6a013ecf..., README.md:14-20) is available: engine vs oracle >= 80 dB per
stem (BASELINE's SDR parity bar is +-0.05 dB, i.e. about 45 dB), and the "directory of reference stems" branch against
wav files written by the oracle (what the Eigen binary's output directory would be)."""
import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def test_sdr_tool_end_to_end_on_synthetic_umxl_weights(pkg, tmp_path):
    spec = importlib.util.spec_from_file_location("sdr_check", HERE / "sdr_check.py")
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    model = str(tmp_path / "ggml-model-synth-umxl-u8.bin.gz")
    pkg.ggml.write_model(model, pkg.ggml.synth_weights(1024, seed=5), 1024)
    wav = HERE / "golden" / "gspi_stereo.wav"
    lines = []
    first = tool.run(model, str(wav), write_oracle_stems=str(tmp_path / "ref"), out=lines.append)
    assert min(first["engine_vs_oracle"]) >= 80.0, lines
    second = tool.run(model, str(wav), stems_dir=str(tmp_path / "ref"), out=lines.append)
    # float32 wav files hold the oracle's stems exactly: the two columns agree
    assert second["engine_vs_dir"] is not None and min(second["engine_vs_dir"]) >= 80.0, lines
    for a, b in zip(second["engine_vs_oracle"], second["engine_vs_dir"]):
        assert abs(a - b) < 1e-6
