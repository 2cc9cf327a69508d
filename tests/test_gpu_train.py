"""The training driver runs on the path bench.py measures (missing #3 / #5): `python -m video_vae_amd.train` replays one captured
hipGraph per (batch, frames) shape of the reference's curriculum (train/rl_nonadversarial.py:276-277,287-295,332), writes the periodic
sample clips (:337-343) and runs the per-epoch eval_step loop (:200-208,362-391)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONUNBUFFERED="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "video_vae_amd.train"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("flavour", ["rl", "model"])
def test_train_driver_replays_one_graph_per_curriculum_shape(dev, tmp_path, flavour):
    """Two epochs of the batch <-> frames curriculum (2 x 8 frames, then 1 x 16): each shape runs eagerly once, is captured, and every later
    step is a replay; sample clips of the reconstruction and the original land every 5 steps; two eval batches per epoch."""
    out = _run(["--small", "--size", "32", "--per_device_batch_size", "2", "--max_frames", "8", "--epochs", "2", "--steps", "12", "--flavour", flavour,
                "--log_every", "3", "--sample_every", "5", "--sample_dir", str(tmp_path), "--sample_ext", "npz", "--eval_steps", "2"])
    lines = out.splitlines()
    caps = [l for l in lines if l.startswith("captured the train step")]
    assert len(caps) == 2 and "(2, 8, 32, 32, 3)" in caps[0] and "(1, 16, 32, 32, 3)" in caps[1], out[-3000:]
    for epoch, (b, t) in enumerate([(2, 8), (1, 16)]):
        logged = [l for l in lines if l.startswith(f"Epoch {epoch}, Step")]
        assert logged and "mode = eager" in logged[0] and all("mode = hipgraph" in l for l in logged[1:]), logged
        assert all(f"effective_batch_size = {b}, effective_max_frames = {t}" in l for l in logged)
        assert sum(l.startswith(f"VALIDATION Epoch {epoch}, Step") for l in lines) == 2
        for i in (4, 9):
            for kind in ("latent", "original"):
                f = tmp_path / "train" / f"epoch{epoch}" / f"video_{i}_{kind}.npz"
                assert f.exists(), (f, os.listdir(tmp_path))
                frames = np.load(f)["frames"]
                assert frames.dtype == np.uint8 and frames.shape[1:] == (32, 32, 3) and 1 <= frames.shape[0] <= t
    # the loss is finite and logged with the reference's keys in every line
    keys = ("MSE", "kl_loss", "selection_loss") + (("rl_loss", "per_sample_MAE") if flavour == "rl" else ())
    for l in lines:
        if l.startswith("Epoch "):
            assert all(k + " = " in l for k in keys) and "nan" not in l.lower(), l
    summ = [l for l in lines if l.startswith("train summary:") and "hipgraph" in l]
    assert summ, out[-2000:]


def test_train_driver_eager_flag_never_captures(dev):
    out = _run(["--small", "--size", "32", "--per_device_batch_size", "2", "--max_frames", "8", "--steps", "6", "--eager", "--log_every", "2"])
    assert "captured the train step" not in out and "mode = hipgraph" not in out and "train summary:" in out
