"""CPU: the host input pipeline (video_vae_amd/data.py) against the reference's batch contract and per-clip recipe
(train/dataloader.py:115-240,387-390; per-rank sharding claude_distributed/dataloader.py:363)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from video_vae_amd import data as D  # noqa: E402


@pytest.fixture(scope="module")
def clips(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("clips"))
    D.write_synthetic_clips(d, 10, 12, 40, 48, seed=1)       # clips 0, 3, 6, 9 have 9 frames, the rest 12
    return d


def test_batch_contract(clips):
    """{"video": float32 (B,T,H,W,3) in [0,1], "mask": float32 (B,T)}; last batch kept unless drop_remainder."""
    dl = D.create_batched_dataloader(clips, batch_size=4, max_frames=8, resize=(32, 32), crop_size=36, num_workers=0)
    bs = list(dl)
    assert [b["video"].shape for b in bs] == [(4, 8, 32, 32, 3), (4, 8, 32, 32, 3), (2, 8, 32, 32, 3)]
    for b in bs:
        assert isinstance(b["video"], np.ndarray) and b["video"].dtype == np.float32 and b["mask"].dtype == np.float32
        assert b["mask"].shape == b["video"].shape[:2]
        assert 0.0 <= float(b["video"].min()) and float(b["video"].max()) <= 1.0
        assert set(np.unique(b["mask"]).tolist()) <= {0.0, 1.0}
    dl = D.create_batched_dataloader(clips, batch_size=4, max_frames=8, resize=(32, 32), crop_size=36, num_workers=0, drop_remainder=True)
    assert len(list(dl)) == 2


def test_short_clips_are_zero_padded_and_masked(clips):
    src = D.VideoDataSource(clips)
    short = [p for p in src.video_paths if np.load(p, mmap_mode="r").shape[0] == 9][0]
    v, m = D.load_video(short, max_frames=16, resize=(20, 20), crop_size=32, rng=np.random.default_rng(0))
    assert v.shape == (16, 20, 20, 3) and v.dtype == np.float32
    assert m.tolist() == [1.0] * 9 + [0.0] * 7
    assert float(np.abs(v[9:]).max()) == 0.0 and float(v[:9].max()) > 0.0


def test_one_crop_for_all_frames_and_values_are_u8_over_255(clips, tmp_path):
    """A clip whose frames are all the same image: every loaded frame must be the same crop of it, values k/255."""
    img = np.random.default_rng(3).integers(0, 256, size=(50, 60, 3), dtype=np.uint8)
    d = tmp_path / "videos0"
    d.mkdir()
    np.save(d / "same.npy", np.repeat(img[None], 7, 0))
    v, m = D.load_video(str(d / "same.npy"), max_frames=5, resize=None, crop_size=32, rng=np.random.default_rng(5))
    assert v.shape == (5, 32, 32, 3) and m.tolist() == [1.0] * 5
    assert all(np.array_equal(v[0], v[i]) for i in range(5))
    u = np.round(v * 255.0)
    assert np.allclose(v, u / 255.0) and u.max() <= 255
    # the crop really is a window of the image
    found = any(np.array_equal(u[0].astype(np.uint8), img[a:a + 32, b:b + 32]) for a in range(50 - 32 + 1) for b in range(60 - 32 + 1))
    assert found


def test_small_frames_are_scaled_up_before_the_crop():
    h, w, sh, sw = D.get_random_crop_params(20, 40, 32, np.random.default_rng(0))       # train/dataloader.py:121-124
    assert (h, w) == (32, 64) and sh == 0 and 0 <= sw <= 32


def test_unreadable_clip_yields_zeros_and_all_ones_mask(tmp_path):
    """The reference's fallback (train/dataloader.py:235-239)."""
    bad = tmp_path / "broken.npy"
    bad.write_bytes(b"not a numpy file")
    v, m = D.load_video(str(bad), max_frames=4, resize=(8, 8), crop_size=16, rng=np.random.default_rng(0))
    assert v.shape == (4, 8, 8, 3) and float(np.abs(v).max()) == 0.0 and m.tolist() == [1.0] * 4


def test_worker_count_does_not_change_the_stream_and_ranks_differ(clips):
    """Crop / window randomness hangs off (seed, epoch, clip), so 0 or 2 worker processes give the same batches; rank r shuffles
    with seed + r (claude_distributed/dataloader.py:363) and sees a different order."""
    kw = dict(batch_size=4, max_frames=8, resize=(32, 32), crop_size=36, shuffle=True, seed=42)
    a = list(D.create_batched_dataloader(clips, num_workers=0, **kw))
    b = list(D.create_batched_dataloader(clips, num_workers=2, prefetch_size=4, **kw))
    assert len(a) == len(b) and all(np.array_equal(x["video"], y["video"]) and np.array_equal(x["mask"], y["mask"]) for x, y in zip(a, b))
    c = list(D.create_batched_dataloader(clips, num_workers=0, rank=1, **kw))
    assert any(not np.array_equal(x["video"], y["video"]) for x, y in zip(a, c))
    # every clip appears exactly once per epoch on a rank
    src = D.VideoDataSource(clips)
    order = [i for _, i in D._EpochSampler(len(src), True, 42, 1)]
    assert sorted(order) == list(range(len(src)))


def test_uint8_hand_over_matches_the_float_contract(clips):
    """as_uint8=True (what DevicePrefetcher moves over PCIe) carries the same pixels: u8 / 255 == the contract's float32 video."""
    kw = dict(batch_size=4, max_frames=8, resize=(32, 32), crop_size=36, shuffle=True, seed=7, num_workers=0)
    f = list(D.create_batched_dataloader(clips, **kw))
    u = list(D.create_batched_dataloader(clips, as_uint8=True, **kw))
    for x, y in zip(f, u):
        assert y["video"].dtype == torch.uint8
        assert np.array_equal(x["video"], y["video"].numpy().astype(np.float32) / 255.0)
        assert np.array_equal(x["mask"], y["mask"].numpy())


def test_unbatched_loader_and_batch_to_video_round_trip(tmp_path):
    """create_dataloader (train/dataloader.py:293-331) yields single clips; batch_to_video (:10-93) drops padded frames, scales to uint8
    and -- as a .npy frame container -- is read back by the loader bit for bit."""
    from video_vae_amd import data as D
    D.write_synthetic_clips(str(tmp_path), 3, 6, 24, 24, seed=1)               # clip 0 has 5 frames, clips 1 and 2 have 6
    items = list(D.create_dataloader(str(tmp_path), max_frames=8, crop_size=16, shuffle=False, num_workers=0))
    assert len(items) == 3
    for it in items:
        assert it["video"].shape == (8, 16, 16, 3) and it["video"].dtype == np.float32 and it["mask"].shape == (8,)
        assert it["mask"].sum() in (5, 6) and float(it["video"].max()) <= 1.0
    out = str(tmp_path / "clip.npy")
    D.batch_to_video({"video": np.stack([it["video"] for it in items]), "mask": np.stack([it["mask"] for it in items])}, out, sample_idx=1)
    back = np.load(out)
    assert back.dtype == np.uint8 and back.shape == (6, 16, 16, 3)                       # padded frames dropped
    assert np.array_equal(back, (np.clip(items[1]["video"][:6], 0, 1) * 255).astype(np.uint8))
    v, m = D.load_video(out, max_frames=6, crop_size=16, rng=np.random.default_rng(0))
    assert np.array_equal((v * 255).round().astype(np.uint8), back) and m.sum() == 6
    with pytest.raises(ValueError):
        D.batch_to_video({"video": items[0]["video"], "mask": np.zeros(8)}, out)
    import shutil
    if not shutil.which("ffmpeg"):
        with pytest.raises(RuntimeError):
            D.batch_to_video({"video": items[0]["video"], "mask": items[0]["mask"]}, str(tmp_path / "clip.mp4"))
    f = np.arange(24 * 32 * 3, dtype=np.uint8).reshape(24, 32, 3)
    assert D.apply_crop(f, 16, (24, 32, 2, 5)).shape == (16, 16, 3) and np.array_equal(D.apply_crop(f, 16, (24, 32, 2, 5)), f[2:18, 5:21])
