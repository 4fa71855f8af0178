"""sfgs.video.FrameDownloader: frames come back bit-exact, in order, through a ring of pinned buffers
without a per-frame host sync."""
import numpy as np
import pytest
import torch


def test_rejects_cpu_tensors_and_bad_depth():
    from sfgs.video import FrameDownloader
    with pytest.raises(ValueError):
        FrameDownloader(depth=0, device="cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 3])
def test_frames_round_trip_in_order(depth):
    from sfgs.video import FrameDownloader
    dev = torch.device("cuda:0")
    dl = FrameDownloader(depth=depth, device=dev)
    with pytest.raises(ValueError):
        dl.submit(torch.zeros(3))
    g = torch.Generator().manual_seed(0)
    frames = [torch.rand(3, 270, 480, generator=g) for _ in range(7)] + [torch.rand(1, 64, 64, generator=g)]  # shape change
    got = []
    for f in frames:
        d = f.to(dev)
        d = d * 1.0                                   # produced by a kernel on the current stream
        got.extend(a.copy() for a in dl.submit(d))
        del d                                         # the downloader keeps what it needs alive
    got.extend(a.copy() for a in dl.drain())
    assert len(got) == len(frames)
    for a, f in zip(got, frames):
        assert a.shape == tuple(f.shape) and np.array_equal(a, f.numpy())
    assert dl.drain() == []
