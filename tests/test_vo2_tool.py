"""tools/test_vo2.py (evaluation_scripts/test_vo2.py, BASELINE.json configs[0] = S-1: 2 frames, 2 edges, 47x156 maps,
num_steps=15, fixedp=2): file hand-off on the CPU at a reduced size, and the full-size clip on the GPU against the same
unroll with the CPU oracle as the correlation lookup."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_pair_driver_writes_the_handoff_files_cpu(tmp_path):
    import pvo_amd.droid_net as dn
    import test_vo2 as T
    from test_droidnet import _OracleCorrBlock
    old = dn.CorrBlock
    dn.CorrBlock = _OracleCorrBlock
    try:
        T.main(["--device", "cpu", "--image_size", "64", "96", "--pairs", "2", "--num_steps", "2", "--scene", "Scene02",
                "--full_flow_dir", str(tmp_path / "full_flow"), "--depth_dir", str(tmp_path / "depth")])
    finally:
        dn.CorrBlock = old
    flows = sorted(os.listdir(tmp_path / "full_flow")); depths = sorted(os.listdir(tmp_path / "depth"))
    assert flows == ["0002_00000.npy", "0002_00001.npy"]
    assert depths == ["0002_00000.npy", "0002_00001.npy", "0002_00002.npy"]           # + the last clip's second frame (:146-150)
    f = np.load(tmp_path / "full_flow" / flows[0]); d = np.load(tmp_path / "depth" / depths[0])
    assert f.shape == (1242, 375, 2) and f.dtype == np.float32                        # cv2.resize(flow, (375, 1242)) of :136
    assert d.shape == (64, 96) and d.dtype == np.float32 and np.isfinite(d).all() and np.isfinite(f).all()


@pytest.mark.gpu
def test_s1_clip_full_size_hip_lookup_against_oracle_lookup():
    """S-1 at full size: 376x1248 images -> 47x156 maps, 2 edges, fixedp=2 (depth-only BA).  The unroll on the GPU (HIP volume
    build and 4-level lookup, fp32) against the identical unroll on the CPU with the C oracle as the lookup; 4 update steps
    keep the CPU side to seconds, every step's flow and depth is compared."""
    import pvo_amd.droid_net as dn
    import test_vo2 as T
    from pvo_amd.synthetic import TrainClips
    from test_droidnet import _OracleCorrBlock
    item = [x[None] for x in TrainClips(2, (376, 1248), length=1, seed=5, step=0.03)[0]]
    images, poses, disps, intr, gt_masks, gt_vals, _ = item
    torch.manual_seed(0)
    net = dn.DroidNet().eval()
    old = dn.CorrBlock
    dn.CorrBlock = _OracleCorrBlock
    try:
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        ref = T.estimate_clip(net, images, poses, intr, gt_vals, num_steps=4)
    finally:
        dn.CorrBlock = old
    dev = torch.device("cuda:0")
    got = T.estimate_clip(net.to(dev), images.to(dev), poses.to(dev), intr.to(dev), gt_vals.to(dev), num_steps=4)
    assert got["full_flow"].shape == (376, 1248, 2) and got["disps"].shape == (2, 376, 1248)
    assert torch.equal(got["poses"].cpu(), poses[0])                                   # fixedp = 2: both poses stay put
    # flow EPE and depth within the tolerance of fp32 convolutions on two devices (the lookup itself is bit-exact)
    epe = (got["full_flow"].cpu() - ref["full_flow"]).norm(dim=-1)
    assert epe.mean() < 2e-3 and epe.max() < 5e-2, (float(epe.mean()), float(epe.max()))
    dd = (got["disps"].cpu() - ref["disps"]).abs()
    assert dd.mean() < 1e-4 and dd.max() < 2e-2, (float(dd.mean()), float(dd.max()))
    assert (ref["disps"] - 1.0).abs().mean() > 1e-3                                    # the depth-only solve did move the depth
