"""tools/train.py (the reference's VO_Module/train.py): the DDP plumbing on two gloo ranks (CPU), and one full-size S-T step
(BASELINE.json configs[4]: 6 frames, 200x400 crop -> 25x50 maps, 15 unrolled updates, bf16 volume) on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _cpu_worker(rank, argv, report):
    import pvo_amd.droid_net as dn
    from test_droidnet import _TorchCorrBlock
    import train as T
    torch.set_num_threads(2)
    dn.CorrBlock = _TorchCorrBlock                     # (the HIP lookup has no CPU fallback: a grid_sample stand-in drives the plumbing)
    T.train(rank, T.parse_args(argv), report)


@pytest.mark.parametrize("mode", ["semisup", "sup"])
def test_train_driver_two_gloo_ranks(tmp_path, mode):
    """two ranks, different clips (DistributedSampler), 3 optimizer steps with restarts: both ranks finish with identical
    weights that differ from the initial ones, the log carries the mode's metric names, and the checkpoint loads into a
    fresh DroidNet through the DDP key prefix the reference's scripts strip (test_vo2.py:83)"""
    import torch.multiprocessing as mp
    argv = ["--gpus", "0,1", "--device", "cpu", "--steps", "3", "--iters", "2", "--n_frames", "4", "--edges", "10", "--crop_size", "64", "96",
            "--mode", mode, "--log_every", "1", "--out_dir", str(tmp_path), "--port", "29541" if mode == "semisup" else "29542", "--restart_prob", "0.5"]
    mgr = mp.Manager()
    report = mgr.dict()
    mp.spawn(_cpu_worker, args=(argv, report), nprocs=2, join=True)
    r0, r1 = report[0], report[1]
    assert r0["steps"] == r1["steps"] == 3
    assert r0["w0"] == r1["w0"]                                            # replicas stay bit-identical
    torch.manual_seed(0)
    from pvo_amd.droid_net import DroidNet
    fresh = DroidNet()
    assert abs(float(next(fresh.parameters()).detach().double().sum()) - r0["w0"]) > 1e-9       # ... and have moved
    names = set(r0["history"][-1][1])
    want = {"residual", "loss", "ph_error"} | ({"ph_cam_error", "gt_mask_error"} if mode == "semisup" else {"rot_error", "f_error", "gt_mask_error"})
    assert want <= names, names
    sd = torch.load(os.path.join(str(tmp_path), "vkitti2_dy_train_final.pth"))
    fresh.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})


@pytest.mark.gpu
def test_full_size_training_step_bf16_volume_against_fp32_volume():
    """S-T: 6 frames at 200x400 (25x50 maps), the 20-edge co-visibility graph, 15 unrolled updates, semi-supervised
    objective of train.py (photometric + mask + residual terms), bf16 correlation volume with the HIP lookup forward and
    backward against the same step with an fp32 volume: loss, final poses / depths, gradient direction per module group."""
    import train as T
    from pvo_amd.droid_net import DroidNet
    from pvo_amd.geom import losses as L
    from pvo_amd.geom.graph_utils import build_frame_graph
    from pvo_amd.geom.se3 import SE3
    from pvo_amd.synthetic import TrainClips
    dev = torch.device("cuda:0")
    args = T.parse_args(["--device", "cuda"])
    item = [x[None].to(dev) for x in TrainClips(6, (200, 400))[3]]
    images, poses, disps, intr, gt_masks, gt_vals, segments = item
    graph = build_frame_graph(poses, disps, intr, num=20, need_inv=False)
    assert 18 <= sum(len(v) for v in graph.values()) <= 20
    outs = {}
    for name, cd in (("fp32", None), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)
        net = DroidNet().to(dev).train()
        Ps = SE3(poses)
        Gs = SE3.IdentityLike(Ps)
        Gs.data[:, 0] = Ps.data[:, 0]; Gs.data[:, 1:] = Ps.data[:, [1]]
        out = net(Gs, images, torch.ones_like(disps[:, :, 3::8, 3::8]), intr / 8.0, graph, num_steps=15, fixedp=2, ret_flow=True,
                  downsample=True, segments=segments, corr_dtype=cd)
        loss, metrics = T.objective(args, L, out, (images, Ps, disps, intr, gt_masks, gt_vals), graph, L.SSIM().to(dev), 0)
        loss.backward()
        torch.cuda.synchronize()
        outs[name] = (float(loss), out[0][-1].data.detach(), out[1][-1].detach(), {k: p.grad for k, p in net.named_parameters()}, metrics)
    (l32, G32, d32, g32, m32), (l16, G16, d16, g16, _) = outs["fp32"], outs["bf16"]
    assert torch.isfinite(torch.tensor(l32)) and abs(l16 - l32) < 3e-2 * abs(l32), (l16, l32)
    assert (G16 - G32).abs().max() < 2e-2 and (d16 - d32).abs().mean() < 5e-3
    assert all(g is not None and torch.isfinite(g).all() for g in g16.values()) and len(g16) == 110
    for group in ("fnet.", "cnet.", "update.gru.", "update.corr_encoder.", "update.flow_encoder."):
        a = torch.cat([g16[k].flatten() for k in sorted(g16) if k.startswith(group)])
        b = torch.cat([g32[k].flatten() for k in sorted(g32) if k.startswith(group)])
        cos = torch.dot(a, b) / (a.norm() * b.norm())
        assert cos > 0.95, (group, float(cos))
    assert {"residual", "ph_cam_error", "gt_mask_error", "ph_error"} <= set(m32)


def _gpu_worker(rank, argv, report):
    import train as T
    T.train(rank, T.parse_args(argv), report)


@pytest.mark.gpu
def test_train_driver_two_ranks_hip_training_step_under_ddp(tmp_path):
    """tools/train.py with TWO ranks running the HIP training step (lookup forward / backward kernels, bf16 volume) under
    DistributedDataParallel: both ranks share cuda:0 over gloo (an RCCL group needs one device per rank; the collective's
    transport is not what is under test) - different clips per rank, two optimizer steps, identical replicas at the end."""
    import torch.multiprocessing as mp
    argv = ["--gpus", "0,0", "--device", "cuda", "--dist_backend", "gloo", "--steps", "2", "--iters", "3", "--n_frames", "4", "--edges", "10",
            "--crop_size", "128", "192", "--log_every", "1", "--out_dir", str(tmp_path), "--port", "29543", "--restart_prob", "0.0"]
    mgr = mp.get_context("spawn").Manager()
    report = mgr.dict()
    mp.spawn(_gpu_worker, args=(argv, report), nprocs=2, join=True)
    r0, r1 = report[0], report[1]
    assert r0["steps"] == r1["steps"] == 2 and r0["w0"] == r1["w0"]
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"]) and r0["loss"] != r1["loss"]          # different clips
    assert os.path.exists(os.path.join(str(tmp_path), "vkitti2_dy_train_final.pth"))
