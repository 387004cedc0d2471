"""The C++17 multi-GPU track driver (include/umx_mgpu.h, host/mgpu.cpp): BASELINE config 4.  On the one-GPU test box:
world 1 through the device-resident path (same kernels, no communicator) must equal the single-GPU whole-track driver
bit for bit; world 2 with both ranks on the one device exercises the RCCL send / recv path where the RCCL build
accepts two ranks per device (otherwise the reason is reported and the test is skipped -- the driver's 8-GPU run is
the one that times it)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_world1_device_path_equals_whole_track_driver(pkg, tmp_path):
    H, N = 1024, 24 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=71), H, compress=False)
    eng = pkg.Engine.from_file(path, N)
    mg = pkg.MultiGpuTrack(eng)
    for L, off in ((int(N * 3.3), None), (int(N * 2.1), 4033), (N // 2, 20000)):
        wave = pkg.ggml.synth_audio(L, 800 + L % 7)
        ref = eng.separate(wave, shift_offset=off)
        got = mg.separate(wave, shift_offset=off)
        for t in range(4):
            assert (got[t] == ref[t]).all(), (L, off, t)
    mg.close()
    eng.close()


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_path_ranks_on_one_device(pkg, model_small, tmp_path, world):
    path, om, targets = model_small
    N, L, seed = 16 * 1024, int(16 * 1024 * 4.6), 811
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), str(ROOT / "tests" / "mgpu_worker.py"), path, str(tmp_path), str(N), str(L), str(seed), "4033"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    if (tmp_path / "mgpu_error.txt").exists():
        pytest.skip("RCCL refused several ranks on one device: " + (tmp_path / "mgpu_error.txt").read_text())
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "mgpu.npy")
    eng = pkg.Engine(targets, 128, N)
    one = np.stack(eng.separate(pkg.ggml.synth_audio(L, seed), shift_offset=4033))
    eng.close()
    assert got.shape == one.shape
    assert (got == one).all()
