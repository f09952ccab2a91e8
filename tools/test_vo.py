"""Run the VO system over an image sequence and evaluate the trajectory - the role of the reference's
evaluation_scripts/test_vo.py (same arguments, VKITTI2 directory layout, KITTI-format output, Sim(3)-aligned ATE).

    python tools/test_vo.py --datapath <.../SceneXX> --weights <checkpoint.pth> [--segm_filter True]

Differences: images are decoded and resized with PIL (no OpenCV in this image; bilinear, like cv2.resize's default),
panoptic ids are decoded in place of panopticapi.rgb2id (id = R + 256 G + 65536 B), and the ATE comes from
pvo_amd.trajectory (Umeyama alignment with scale, translation part - what evo's APE call at test_vo.py:162-163 computes).
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SPLIT = {"train": "clone", "val": "15-deg-left", "test": "30-deg-right"}
VKITTI2_INTRINSICS = (725.0087, 725.0087, 620.5, 187.0)            # test_vo.py:21


def rgb2id(color):
    c = np.asarray(color, dtype=np.int64)
    return c[..., 0] + 256 * c[..., 1] + 65536 * c[..., 2]


def image_stream(datapath, image_size=(240, 808), mode="val", segm_filter=False):
    """yields (t, image [3,H,W] int BGR, intrinsics [4], segm [1,1,H/8,W/8] int or None)  (test_vo.py:19-56)"""
    from PIL import Image
    fx, fy, cx, cy = VKITTI2_INTRINSICS
    images = sorted(glob.glob(os.path.join(datapath, SPLIT[mode], "frames/rgb/Camera_0/*.jpg")))
    segms = sorted(glob.glob(os.path.join(datapath, SPLIT[mode], "panFPN_segm/*.png")))
    h1, w1 = int(image_size[0]), int(image_size[1])
    for t, path in enumerate(images):
        im = Image.open(path).convert("RGB")
        w0, h0 = im.size
        rgb = np.asarray(im.resize((w1, h1), Image.BILINEAR))
        rgb = rgb[:h1 - h1 % 8, :w1 - w1 % 8]
        image = torch.as_tensor(rgb[..., ::-1].copy()).int().permute(2, 0, 1)          # BGR, as cv2.imread returns; int32 as test_vo.py:41
        segm = None
        if segm_filter:
            ids = rgb2id(np.asarray(Image.open(segms[t]).convert("RGB")))
            s = torch.as_tensor(ids).float()[None, None]
            s = F.interpolate(s, size=(h1, w1))[..., :h1 - h1 % 8, :w1 - w1 % 8]
            segm = F.interpolate(s, scale_factor=1 / 8, recompute_scale_factor=True).int()
        intr = torch.as_tensor([fx, fy, cx, cy])
        intr[0:2] *= w1 / w0
        intr[2:4] *= h1 / h0
        yield t, image, intr, segm


def read_vkitti2_poses(path):
    """camera-to-world 4x4 poses of Camera_0 from a VKITTI2 extrinsic.txt (test_vo.py:127-148: every second row,
    world-to-camera matrices, inverted)"""
    raw = np.loadtxt(path, delimiter=" ", skiprows=1)[::2, 2:]
    if raw.shape[1] != 16:
        raise ValueError("Virtual KITTI 2 pose files have 16 matrix entries per row")
    return np.stack([np.linalg.inv(r.reshape(4, 4)) for r in raw.astype(float)])


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--datapath")
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--weights", default=None)
    p.add_argument("--buffer", type=int, default=1024)
    p.add_argument("--image_size", default=[240, 808])
    p.add_argument("--disable_vis", action="store_true")
    p.add_argument("--use_aff_bri", type=bool, default=False)
    p.add_argument("--beta", type=float, default=0.6)
    p.add_argument("--filter_thresh", type=float, default=1.75)
    p.add_argument("--warmup", type=int, default=12)
    p.add_argument("--keyframe_thresh", type=float, default=2.25)
    p.add_argument("--frontend_thresh", type=float, default=12.0)
    p.add_argument("--frontend_window", type=int, default=25)
    p.add_argument("--frontend_radius", type=int, default=2)
    p.add_argument("--frontend_nms", type=int, default=1)
    p.add_argument("--backend_thresh", type=float, default=15.0)
    p.add_argument("--backend_radius", type=int, default=2)
    p.add_argument("--backend_nms", type=int, default=3)
    p.add_argument("--segm_filter", type=bool, default=False)
    p.add_argument("--thresh", type=float, default=0.8)
    p.add_argument("--out", default="shared_data/traj")
    p.add_argument("--pipelined", action="store_true", help="MI355X build: run the second half of a keyframe update inside the next track() call "
                   "(same trajectory bit for bit; pvo_amd/droid.py)")
    return p.parse_args(argv)


def main(argv=None):
    from pvo_amd.droid import Droid
    from pvo_amd.handoff import write_kitti_trajectory
    from pvo_amd.trajectory import ate_rmse
    args = parse_args(argv)
    args.half_update = True
    if args.datapath.endswith("20"):
        args.thresh = 0.9                                             # test_vo.py:94-95
    droid = Droid(args)
    for t, image, intr, segm in image_stream(args.datapath, args.image_size, "val", args.segm_filter):
        droid.track(t, image, intrinsics=intr, segments=segm)
    print("video frames:", droid.video.counter)
    traj = droid.terminate(image_stream(args.datapath, args.image_size, "val", args.segm_filter), need_inv=True)
    out_dir = os.path.join(args.out, os.path.basename(args.datapath.rstrip("/")), SPLIT["val"])
    est_file = os.path.join(out_dir, "pvo_traj.txt")
    write_kitti_trajectory(est_file, traj)
    print("trajectory written to", est_file)
    gt_file = os.path.join(args.datapath, SPLIT["val"], "extrinsic.txt")
    if os.path.exists(gt_file):
        gt = read_vkitti2_poses(gt_file)[:, :3, 3]
        n = min(len(gt), len(traj))
        print("ATE-RMSE (Sim(3)-aligned, translation): %.4f m over %d poses" % (ate_rmse(traj[:n, :3], gt[:n]), n))


if __name__ == "__main__":
    main()
