"""Differentiable dense bundle adjustment in PyTorch (training path / any-device fallback).

Mirror of the reference's ``geom/ba.py`` (VO_Module/droid_slam/geom/ba.py:31-157): `BA` and `MoBA`
keep the reference's signature and semantics - weights `0.001 * valid * weight` (:45), depth
diagonal `C + eta + 1e-7` (:91), poses `< fixedp` held fixed (:77-79), retraction on the left,
`disps > 10 -> 0` then `clamp(min=0)` (:103-104).  This is BASELINE config 1's Python BA and the
one used inside the training loop (droid_net.py:341); inference uses the HIP solver
(`droid_backends.ba`), whose normal equations coincide with these when all depths exceed 0.25.

Construction differs from the reference: normal-equation blocks come from einsum contractions
over the 2*HW residual rows (no transposed weighted copies of the Jacobians), the block scatter is
`index_add_` on flattened block ids, and the solve is `chol.schur_solve`'s block contraction.
"""
import torch

from . import projective_ops as pops
from .chol import block_solve, schur_solve


def _scatter_blocks(src, row, col, n, m):
    """Sum src[:, e] into block (row[e], col[e]) of an [n, m] grid; out-of-range ids are dropped
    (ba.py:12-14 `safe_scatter_add_mat`)."""
    keep = (row >= 0) & (col >= 0) & (row < n) & (col < m)
    out = src.new_zeros((src.shape[0], n * m) + tuple(src.shape[2:]))
    return out.index_add_(1, (row[keep] * m + col[keep]).to(src.device), src[:, keep.to(src.device)])


def _scatter_rows(src, row, n):
    """ba.py:16-18 `safe_scatter_add_vec`."""
    keep = (row >= 0) & (row < n)
    out = src.new_zeros((src.shape[0], n) + tuple(src.shape[2:]))
    return out.index_add_(1, row[keep].to(src.device), src[:, keep.to(src.device)])


def disp_retr(disps, dz, ii):
    """disps[:, ii] += dz (ba.py:21-23)."""
    return disps + _scatter_rows(dz, ii, disps.shape[1])


def pose_retr(poses, dx, ii):
    """poses[:, ii] <- Exp(dx) * poses[:, ii] (ba.py:26-28)."""
    return poses.retr(_scatter_rows(dx, ii, poses.shape[1]))


def _pose_system(Ji, Jj, w, r, ii, jj, P):
    """Pose-pose blocks and gradient from per-edge Jacobians [B,N,R,D], weights/residuals [B,N,R]."""
    wJi, wJj = w[..., None] * Ji, w[..., None] * Jj
    blk = lambda a, b: torch.einsum("bnrd,bnre->bnde", a, b)
    H = (_scatter_blocks(blk(wJi, Ji), ii, ii, P, P) + _scatter_blocks(blk(wJi, Jj), ii, jj, P, P) +
         _scatter_blocks(blk(wJj, Ji), jj, ii, P, P) + _scatter_blocks(blk(wJj, Jj), jj, jj, P, P))
    v = (_scatter_rows(torch.einsum("bnrd,bnr->bnd", wJi, r), ii, P) +
         _scatter_rows(torch.einsum("bnrd,bnr->bnd", wJj, r), jj, P))
    D = Ji.shape[-1]
    return H.view(H.shape[0], P, P, D, D), v, wJi, wJj


def _linearise(target, weight, poses, disps, intrinsics, ii, jj):
    B, N = target.shape[0], ii.shape[0]
    coords, valid, (Ji, Jj, Jz) = pops.projective_transform(poses, disps, intrinsics, ii, jj, jacobian=True)
    D = poses.manifold_dim
    r = (target - coords).reshape(B, N, -1)
    w = 0.001 * (valid * weight).reshape(B, N, -1)
    return r, w, Ji.reshape(B, N, -1, D), Jj.reshape(B, N, -1, D), Jz


def BA(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp=1, rig=1):
    """One Gauss-Newton step over poses and inverse depths (ba.py:31-106)."""
    B, P, ht, wd = disps.shape
    N, HW = ii.shape[0], ht * wd
    r, w, Ji, Jj, Jz = _linearise(target, weight, poses, disps, intrinsics, ii, jj)

    kx, kk = torch.unique(ii, return_inverse=True)
    M = kx.shape[0]
    Pf = P // rig - fixedp
    pi, pj = ii // rig - fixedp, jj // rig - fixedp

    H, v, wJi, wJj = _pose_system(Ji, Jj, w, r, pi, pj, Pf)

    # pose-depth coupling and the (diagonal) depth block, per pixel: contract the two residual rows
    D = Ji.shape[-1]
    Jz2 = Jz.reshape(B, N, HW, 2)
    Ei = torch.einsum("bnhcd,bnhc->bndh", wJi.view(B, N, HW, 2, D), Jz2)
    Ej = torch.einsum("bnhcd,bnhc->bndh", wJj.view(B, N, HW, 2, D), Jz2)
    w2, r2 = w.view(B, N, HW, 2), r.view(B, N, HW, 2)
    wk = (w2 * r2 * Jz2).sum(-1)
    Ck = (w2 * Jz2 * Jz2).sum(-1)

    E = (_scatter_blocks(Ei, pi, kk, Pf, M) + _scatter_blocks(Ej, pj, kk, Pf, M)).view(B, Pf, M, D, HW)
    C = _scatter_rows(Ck, kk, M) + eta.reshape(B, M, HW) + 1e-7
    wz = _scatter_rows(wk, kk, M)

    dx, dz = schur_solve(H, E, C, v, wz)

    poses = pose_retr(poses, dx, torch.arange(Pf) + fixedp)
    disps = disp_retr(disps, dz.view(B, -1, ht, wd), kx)
    disps = torch.where(disps > 10, torch.zeros_like(disps), disps).clamp(min=0.0)
    return poses, disps


def MoBA(target, weight, eta, poses, disps, intrinsics, ii, jj, fixedp=1, rig=1):
    """Motion-only step: depths held fixed (ba.py:109-157)."""
    P = disps.shape[1]
    r, w, Ji, Jj, _ = _linearise(target, weight, poses, disps, intrinsics, ii, jj)
    Pf = P // rig - fixedp
    H, v, _, _ = _pose_system(Ji, Jj, w, r, ii // rig - fixedp, jj // rig - fixedp, Pf)
    dx = block_solve(H, v)
    return pose_retr(poses, dx, torch.arange(Pf) + fixedp)
