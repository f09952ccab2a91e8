"""Trajectory error: Sim(3) (Umeyama) alignment + translation RMSE.

What the reference gets from `evo` (`main_ape.ape(..., pose_relation=translation_part, align=True,
correct_scale=True)`, VO_Module/evaluation_scripts/test_vo.py:162-163); evo is not available in this image."""
import numpy as np


def umeyama(src, dst, with_scale=True):
    """least-squares similarity (s, R, t) with dst ~ s R src + t  (Umeyama 1991); src, dst [N,3]"""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / src.shape[0]
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    var = (xs ** 2).sum() / src.shape[0]
    s = float(np.trace(np.diag(D) @ S) / var) if with_scale else 1.0
    t = mu_d - s * R @ mu_s
    return s, R, t


def ate_rmse(est_xyz, gt_xyz, align=True, correct_scale=True):
    """translation-part absolute trajectory error after alignment (evo APE)"""
    est, gt = np.asarray(est_xyz, np.float64), np.asarray(gt_xyz, np.float64)
    if align:
        s, R, t = umeyama(est, gt, with_scale=correct_scale)
        est = (s * (R @ est.T)).T + t
    return float(np.sqrt(((est - gt) ** 2).sum(1).mean()))


def camera_centres(poses_w2c):
    """positions of the cameras in the world from world-to-camera poses [N,7] (t, q xyzw): c = -R^T t"""
    p = np.asarray(poses_w2c, np.float64)
    t, q = p[:, :3], p[:, 3:]
    qv, w = -q[:, :3], q[:, 3:4]                     # conjugate
    uv = 2.0 * np.cross(qv, t)
    return -(t + w * uv + np.cross(qv, uv))
