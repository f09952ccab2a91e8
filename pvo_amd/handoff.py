"""On-disk hand-off to the video-panoptic-segmentation half of PVO (out of scope here): per image pair one
`.npy` with the full optical flow and one with the inverse depth, as evaluation_scripts/test_vo2.py:131-143
writes them (`<root>/full_flow/<id>.npy`, `<root>/depth/<id>.npy`, float32, written with numpy.save)."""
import os

import numpy as np
import torch
import torch.nn.functional as F


def _resize_bilinear(a, size_hw):
    """half-pixel-centre bilinear resize of an [H,W,C] array (what cv2.resize's default does, test_vo2.py:136)"""
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)[None].float()
    return F.interpolate(t, size=size_hw, mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()


def save_flow_depth(root, img_id, full_flow, disp, valid=None, resize_hw=None):
    """full_flow [H,W,2] (pixels, image resolution), disp [H,W] inverse depth, valid [H,W,1] optional mask.
    Returns the two paths written."""
    flow = full_flow.detach().cpu().numpy() if isinstance(full_flow, torch.Tensor) else np.asarray(full_flow)
    flow = flow.astype(np.float32)
    if valid is not None:
        flow = flow * (valid.detach().cpu().numpy() if isinstance(valid, torch.Tensor) else np.asarray(valid))
    if resize_hw is not None:
        flow = _resize_bilinear(flow, resize_hw)
    depth = (disp.detach().cpu().numpy() if isinstance(disp, torch.Tensor) else np.asarray(disp)).astype(np.float32)
    paths = []
    for sub, arr in (("full_flow", flow), ("depth", depth)):
        d = os.path.join(root, sub)
        os.makedirs(d, exist_ok=True)
        paths.append(os.path.join(d, img_id + ".npy"))
        np.save(paths[-1], arr)
    return paths


def write_kitti_trajectory(path, traj):
    """[N,7] (t, q xyzw) camera-to-world poses -> KITTI format, 12 numbers per line (the file test_vo.py:150-159
    writes through evo's write_kitti_poses_file)."""
    from .geom.se3 import SE3
    T = SE3(torch.as_tensor(np.asarray(traj), dtype=torch.float64)).matrix().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        for m in T:
            f.write(" ".join("%.9e" % v for v in m[:3].reshape(-1)) + "\n")
