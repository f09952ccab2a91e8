"""End-to-end closed loop on a synthetic scene (ATE-RMSE half of BASELINE.json's metric).

The frontend + FactorGraph + dense BA run once on the HIP kernels and once on the CPU oracle (same
host logic, `OracleVideo` below), both driven by ground-truth correspondences (+ fixed noise) in
place of the learned operator.  north_star: ATE-RMSE within 1e-3 of the reference path."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from pvo_amd.frontend import DroidFrontend
from pvo_amd.synthetic import OracleFlowOperator, PlaneScene, run_sequence
from pvo_amd.trajectory import ate_rmse, camera_centres, umeyama


def test_umeyama_recovers_a_known_similarity():
    g = np.random.default_rng(0)
    src = g.standard_normal((40, 3))
    A = g.standard_normal((3, 3)); R, _ = np.linalg.qr(A)
    if np.linalg.det(R) < 0:
        R[:, 0] *= -1
    s, t = 2.5, np.array([0.3, -1.0, 4.0])
    dst = s * src @ R.T + t
    s2, R2, t2 = umeyama(src, dst)
    assert abs(s2 - s) < 1e-9 and np.allclose(R2, R, atol=1e-9) and np.allclose(t2, t, atol=1e-9)
    assert ate_rmse(src, dst) < 1e-9 and ate_rmse(src, dst, align=False) > 1.0
    p = np.array([[1.0, 2.0, 3.0, 0, 0, 0, 1.0]])
    assert np.allclose(camera_centres(p), [[-1, -2, -3]])


class OracleVideo:
    """DepthVideo's interface on CPU tensors, native calls answered by the CPU oracle"""

    def __init__(self, ht, wd, buffer=32):
        self.ht, self.wd, self.counter = ht * 8, wd * 8, 0
        self.poses = torch.zeros(buffer, 7); self.poses[:, 6] = 1
        self.disps = torch.ones(buffer, ht, wd)
        self.intrinsics = torch.zeros(buffer, 4)
        self.tstamp = torch.zeros(buffer); self.dirty = torch.zeros(buffer, dtype=torch.bool)
        z = torch.zeros(buffer, 1, 1, 1)
        self.nets = self.inps = self.fmaps = z
        self.segms = torch.zeros(buffer, 1, ht, wd, dtype=torch.int)
        self.segm_filter, self.thresh = False, 0.8

    def append(self, tstamp, pose, disp, intrinsics, *unused):
        k = self.counter
        if pose is not None:
            self.poses[k] = pose
        self.intrinsics[k] = intrinsics
        self.counter = k + 1

    def reproject(self, ii, jj):
        c, v = O.reproject(self.poses.numpy(), self.disps.numpy(), self.intrinsics.numpy(), np.asarray(ii), np.asarray(jj))
        return torch.from_numpy(c)[None], torch.from_numpy(v)[None]

    def distance(self, ii, jj, beta=0.3, bidirectional=True):
        ii, jj = np.asarray(ii, np.int64).reshape(-1), np.asarray(jj, np.int64).reshape(-1)
        a = (self.poses[:self.counter].numpy().copy(), self.disps.numpy(), self.intrinsics[0].numpy())
        d = O.frame_distance(*a, ii, jj, beta)
        if bidirectional:
            d = 0.5 * (d + O.frame_distance(*a, jj, ii, beta))
        return torch.from_numpy(d)

    def normalize(self):
        n = self.counter
        s = self.disps[:n].mean()
        self.disps[:n] /= s
        self.poses[:n, :3] *= s
        self.dirty[:n] = True

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        r = O.ba(self.poses.numpy(), self.disps.numpy(), self.intrinsics[0].numpy(), target.numpy(), weight.numpy(),
                 eta.numpy(), ii.numpy(), jj.numpy(), t0, t1, itrs, lm, ep, motion_only=motion_only)
        self.poses.copy_(torch.from_numpy(r["poses"])); self.disps.copy_(torch.from_numpy(r["disps"]).clamp(min=0.001))


def _oracle_reproject(poses, disps, intr, ii, jj):
    c, _ = O.reproject(poses.numpy(), disps.numpy(), intr.numpy(), ii.numpy(), jj.numpy())
    return torch.from_numpy(c)


@pytest.mark.gpu
def test_closed_loop_ate_hip_equals_oracle_path(cuda):
    from pvo_amd import droid_backends as db
    from pvo_amd.depth_video import DepthVideo
    scene = PlaneScene(ht=24, wd=32, n_frames=14, seed=0)
    kw = dict(warmup=8, keyframe_thresh=0.5, frontend_thresh=16.0, frontend_window=20, frontend_radius=2, frontend_nms=1)
    # --- HIP path
    video = DepthVideo(image_size=(scene.ht * 8, scene.wd * 8), buffer=32, device=cuda)
    op = OracleFlowOperator(scene, video, lambda p, d, k, i, j: db.reproject(p, d, k, i, j)[0])
    fe = DroidFrontend(op, video, device=cuda, **kw)
    poses_hip, frames_hip = run_sequence(scene, video, fe, op)
    # --- CPU oracle path, identical host logic
    ov = OracleVideo(scene.ht, scene.wd, buffer=32)
    op2 = OracleFlowOperator(scene, ov, _oracle_reproject)
    fe2 = DroidFrontend(op2, ov, device="cpu", **kw)
    fe2.graph.corr_impl = "none"
    fe2.graph.corr = type("NoVolumes", (), {"__call__": lambda self, coords, **kw: None})()   # "a volume exists" (droid_frontend.py:42)
    poses_cpu, frames_cpu = run_sequence(scene, ov, fe2, op2)
    assert frames_hip == frames_cpu and len(frames_hip) >= 10          # same keyframe decisions
    gt = camera_centres(scene.poses[frames_hip].numpy())
    ate_hip = ate_rmse(camera_centres(poses_hip.numpy()), gt)
    ate_cpu = ate_rmse(camera_centres(poses_cpu.numpy()), gt)
    print("ATE-RMSE hip %.6f cpu-oracle %.6f (trajectory length %.2f)" % (ate_hip, ate_cpu, np.linalg.norm(gt[-1] - gt[0])))
    assert abs(ate_hip - ate_cpu) < 1e-3                                 # north_star tolerance
    assert ate_hip < 0.05 * np.linalg.norm(gt[-1] - gt[0])               # and the loop actually tracks the camera
    assert np.abs(camera_centres(poses_hip.numpy()) - camera_centres(poses_cpu.numpy())).max() < 5e-3


@pytest.mark.gpu
def test_closed_loop_at_the_driver_map_size_with_keyframe_removal_and_global_ba(cuda, monkeypatch):
    """The same closed loop at the reference driver's map size (30 x 101 = 240 x 808 / 8: 3030 pixels, 512-pixel Schur chunks) through
    the REAL DroidFrontend (window with its inactive edges, proximity factors, and - every third frame barely moves - the keyframe
    test's removal branch, droid_frontend.py:54-58) and the REAL DroidBackend (two global bundle adjustments, droid.py:84-90), once on
    the HIP kernels and once on the CPU oracle: the same keyframe decisions, removals included; ATE within the north star's 1e-3
    before and after the global BA."""
    from argparse import Namespace
    from pvo_amd import droid_backends as db
    from pvo_amd.backend import DroidBackend
    from pvo_amd.depth_video import DepthVideo
    import pvo_amd.modules.corr as corr_mod
    n = 26
    scene = PlaneScene(ht=30, wd=101, n_frames=n, seed=0, step=0.06, pattern=(1.0, 1.0, 0.15))
    kw = dict(warmup=8, keyframe_thresh=0.6, frontend_thresh=16.0, frontend_window=25, frontend_radius=2, frontend_nms=1)
    bargs = lambda dev: Namespace(device=dev, backend_radius=2, backend_nms=3, backend_thresh=15.0, beta=0.3, backend_corr="alt")
    # --- HIP path
    video = DepthVideo(image_size=(scene.ht * 8, scene.wd * 8), buffer=n + 8, device=cuda)
    op = OracleFlowOperator(scene, video, lambda p, d, k, i, j: db.reproject(p, d, k, i, j)[0])
    fe = DroidFrontend(op, video, device=cuda, **kw)
    be = DroidBackend(Namespace(update=op), video, bargs(str(cuda)))
    hip_before, hip_after, frames_hip = run_sequence(scene, video, fe, op, backend=be, backend_steps=(2, 3))
    # --- CPU oracle path, identical host logic (the stand-in operator reads no correlation features: none are computed there)
    ov = OracleVideo(scene.ht, scene.wd, buffer=n + 8)
    op2 = OracleFlowOperator(scene, ov, _oracle_reproject)
    fe2 = DroidFrontend(op2, ov, device="cpu", **kw)
    fe2.graph.corr_impl = "none"
    fe2.graph.corr = type("NoVolumes", (), {"__call__": lambda self, coords, **kw: None})()
    monkeypatch.setattr(corr_mod, "AltCorrBlock", lambda *a, **k: (lambda *a2, **k2: None))
    be2 = DroidBackend(Namespace(update=op2), ov, bargs("cpu"))
    cpu_before, cpu_after, frames_cpu = run_sequence(scene, ov, fe2, op2, backend=be2, backend_steps=(2, 3))
    monkeypatch.undo()
    assert frames_hip == frames_cpu                                      # same keyframe decisions
    assert fe.keyframes_removed == fe2.keyframes_removed >= 4 and len(frames_hip) >= 16
    gt = camera_centres(scene.poses[frames_hip].numpy())
    length = np.linalg.norm(np.diff(gt, axis=0), axis=1).sum()
    for name, a, b in (("frontend", hip_before, cpu_before), ("global BA", hip_after, cpu_after)):
        ate_h, ate_c = ate_rmse(camera_centres(a.numpy()), gt), ate_rmse(camera_centres(b.numpy()), gt)
        print("%s: ATE-RMSE hip %.6f cpu-oracle %.6f (path length %.2f)" % (name, ate_h, ate_c, length))
        assert abs(ate_h - ate_c) < 1e-3                                 # north_star tolerance
        assert ate_h < 0.05 * length
        assert np.abs(camera_centres(a.numpy()) - camera_centres(b.numpy())).max() < 5e-3
