"""Per-pair flow and depth for the panoptic half of PVO - the role of the reference's
VO_Module/evaluation_scripts/test_vo2.py (second command of tools/test_vo_scene.sh, BASELINE.json configs[0]):

    python tools/test_vo2.py --scene Scene02 [--weights_file checkpoint.pth] [--pairs 4]

For every clip of `n_frames` = 2 frames (test_vo2.py:55-143): images resized to 376x1248, unit inverse depth, the two-frame
graph 0 <-> 1, `DroidNet.forward(num_steps=15, fixedp=2, ret_flow=True, downsample=True)` (HIP correlation volume
build + lookup at 47x156, PyTorch BA with both poses fixed - a depth-only solve), then
    full_flow = upsample_inter(full_flows[-1] * 8)[0, 0]           masked by gt_vals, resized, -> <full_flow_dir>/<id>.npy
    depth     = disps_est[-1][0, 0]                                 -> <depth_dir>/<id>.npy
and the last clip's second depth map under the next id (:146-150).

Data: the VKITTI2 reader is outside this build's scope and no dataset / checkpoint exists in this environment
(datasets/, checkpoints/ hold READMEs): clips come from `pvo_amd.synthetic.TrainClips` at the reference's image size in
the reader's item layout; without --weights_file the network is randomly initialised (seed 0).  cv2.resize(flow, (375,
1242)) of the reference - (width, height) = (375, 1242), i.e. a 1242 x 375 array - is reproduced with a half-pixel-centre
bilinear resize (pvo_amd/handoff.py).
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SCENES = {"Scene01": "0001", "Scene02": "0002", "Scene06": "0006", "Scene18": "0018", "Scene20": "0020"}


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--n_frames", type=int, default=2)
    p.add_argument("--save_npy", type=bool, default=True)
    p.add_argument("--image_size", type=int, nargs=2, default=[376, 1248])
    p.add_argument("--scene", default="Scene02")
    p.add_argument("--full_flow_dir", default="shared_data/full_flow")
    p.add_argument("--depth_dir", default="shared_data/depth")
    p.add_argument("--weights_file", default=None)
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--pairs", type=int, default=4, help="clips to process (the reference runs the whole scene)")
    p.add_argument("--num_steps", type=int, default=15)
    return p.parse_args(argv)


def resize_clip(x, size, last_dim_channels):
    """bilinear align_corners resize of [B,N,C,H,W] or [B,N,H,W,C] (test_vo2.py:24-37)"""
    import torch.nn.functional as F
    if last_dim_channels:
        x = x.permute(0, 1, 4, 2, 3)
    b, n, c, h, w = x.shape
    y = F.interpolate(x.reshape(b * n, c, h, w), size=tuple(size), mode="bilinear", align_corners=True).view(b, n, c, *size)
    return y.permute(0, 1, 3, 4, 2).contiguous() if last_dim_channels else y


def two_frame_graph(n):
    return OrderedDict((i, [j for j in range(n) if abs(i - j) == 1]) for i in range(n))


@torch.no_grad()
def estimate_clip(model, images, poses, intrinsics, gt_vals=None, num_steps=15, passes=1):
    """test_vo2.py:103-127 for one clip: returns dict(full_flow [H,W,2], cam_flow, mask, disps [N,H,W], poses)"""
    from pvo_amd.droid_net import upsample_inter
    from pvo_amd.geom.graph_utils import graph_to_edge_list
    from pvo_amd.geom.projective_ops import coords_grid, projective_transform
    from pvo_amd.geom.se3 import SE3
    n = images.shape[1]
    graph = two_frame_graph(n)
    ii, jj, _ = graph_to_edge_list(graph)
    ii, jj = ii.to(images.device), jj.to(images.device)
    H, W = images.shape[-2:]
    Gs = SE3(poses)
    disp0 = torch.ones(1, n, H // 8, W // 8, device=images.device)
    for _ in range(passes):
        poses_est, disps_est, residuals, full_flows, masks = model(Gs, images, disp0, intrinsics / 8, graph, num_steps=num_steps,
                                                                   fixedp=2, ret_flow=True, downsample=True)
        Gs, disp0 = poses_est[-1], disps_est[-1][:, :, 3::8, 3::8]
    coords1, _ = projective_transform(poses_est[-1], disps_est[-1], intrinsics, ii, jj)
    cam_flow = (coords1 - coords_grid(H, W, device=images.device))[0, 0]
    full_flow = upsample_inter(full_flows[-1] * 8)[0, 0]
    mask = (masks[-1][0, 0].mean(-1, keepdim=True) >= 0.5).float()
    return dict(full_flow=full_flow, cam_flow=cam_flow, resd=full_flow - cam_flow, mask=mask, disps=disps_est[-1][0],
                poses=poses_est[-1].data[0], residual=residuals[-1])


def main(argv=None):
    from pvo_amd.droid_net import DroidNet
    from pvo_amd.handoff import save_flow_depth
    from pvo_amd.synthetic import TrainClips
    args = parse_args(argv)
    dev = torch.device(args.device)
    torch.manual_seed(0)
    model = DroidNet()
    if args.weights_file:
        sd = torch.load(args.weights_file, map_location="cpu")
        model.load_state_dict(OrderedDict((k.replace("module.", ""), v) for k, v in sd.items()))      # test_vo2.py:83-84
    model.to(dev).eval()
    H, W = args.image_size
    clips = TrainClips(args.n_frames, (H, W), length=args.pairs, seed=sum(map(ord, args.scene)), step=0.03)
    root = os.path.dirname(os.path.abspath(args.full_flow_dir)) if os.path.basename(args.full_flow_dir) == "full_flow" else None
    out, img_id, t0 = None, None, time.perf_counter()
    for k in range(args.pairs):
        images, poses, disps, intr, gt_masks, gt_vals, _ = [x[None].to(dev) for x in clips[k]]
        out = estimate_clip(model, images, poses, intr, gt_vals, args.num_steps)
        img_id = "%s_%05d" % (SCENES.get(args.scene, "0000"), k)
        print(img_id)
        if args.save_npy:
            if root is not None and os.path.basename(args.depth_dir) == "depth":
                save_flow_depth(root, img_id, out["full_flow"], out["disps"][0], valid=gt_vals[0, 0], resize_hw=(1242, 375))
            else:
                os.makedirs(args.full_flow_dir, exist_ok=True); os.makedirs(args.depth_dir, exist_ok=True)
                from pvo_amd.handoff import _resize_bilinear
                np.save(os.path.join(args.full_flow_dir, img_id + ".npy"),
                        _resize_bilinear((out["full_flow"] * gt_vals[0, 0]).cpu().numpy().astype(np.float32), (1242, 375)))
                np.save(os.path.join(args.depth_dir, img_id + ".npy"), out["disps"][0].cpu().numpy().astype(np.float32))
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    print("Finished building %d img with flows (%.2f s per pair)" % (args.pairs, (time.perf_counter() - t0) / max(args.pairs, 1)))
    if out is not None and args.save_npy:                               # test_vo2.py:146-150: the last clip's second depth map
        nxt = "%s_%05d" % (img_id.rsplit("_", 1)[0], int(img_id.rsplit("_", 1)[1]) + 1)
        os.makedirs(args.depth_dir, exist_ok=True)
        np.save(os.path.join(args.depth_dir, nxt + ".npy"), out["disps"][1].cpu().numpy().astype(np.float32))
        print(nxt)


if __name__ == "__main__":
    main()
