"""GPU: the device end of the host input pipeline (DevicePrefetcher): pinned uint8 -> H2D on a side stream -> /255 on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_prefetcher_delivers_the_contract_batches(dev, tmp_path):
    """Device batches equal the host contract's float32 batches bit for bit (u8 / 255 is the same IEEE division on either side),
    arrive in order, and a train step can consume them."""
    import video_vae_amd as V
    from video_vae_amd import data as D, loss as L, optim
    d = str(tmp_path)
    D.write_synthetic_clips(d, 9, 10, 40, 40, seed=2)
    kw = dict(batch_size=2, max_frames=8, resize=(32, 32), crop_size=36, shuffle=True, seed=11, num_workers=2, prefetch_size=4,
              drop_remainder=True)
    want = list(D.create_batched_dataloader(d, **kw))
    got = []
    for b in D.DevicePrefetcher(D.create_batched_dataloader(d, as_uint8=True, **kw), dev, dtype=torch.float32):
        assert b["video"].is_cuda and b["video"].dtype == torch.float32 and b["mask"].dtype == torch.float32
        got.append({k: v.cpu().numpy() for k, v in b.items()})
    assert len(got) == len(want) == 4
    for g, w in zip(got, want):
        assert np.array_equal(g["video"], w["video"]) and np.array_equal(g["mask"], w["mask"])
    # the float32 host contract is accepted too, and the compute dtype can be asked for directly
    b16 = next(iter(D.DevicePrefetcher(iter(want), dev, dtype=torch.bfloat16)))
    assert b16["video"].dtype == torch.bfloat16
    assert torch.equal(b16["video"].cpu(), torch.from_numpy(want[0]["video"]).to(torch.bfloat16))
    tiny = dict(height=32, width=32, channels=3, patch_size=8, encoder_depth=1, decoder_depth=1, mlp_dim=64, num_heads=4,
                qkv_features=32, max_temporal_len=8, spatial_compression_rate=4, unembedding_upsample_rate=4)
    m = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.float32, **tiny).to(dev)
    opt = optim.Optimizer(m, 1e-3)
    for b in D.DevicePrefetcher(D.create_batched_dataloader(d, as_uint8=True, **kw), dev):
        loss, _ = L.train_step(m, opt, b["video"], b["mask"], L.HPARAMS, 16, V.Rngs(3))
        assert torch.isfinite(loss)
