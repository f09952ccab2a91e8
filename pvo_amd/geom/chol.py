"""Damped normal-equation solvers for the differentiable (PyTorch) bundle adjustment.

Mirror of the reference's ``geom/chol.py`` (VO_Module/droid_slam/geom/chol.py:5-73): same entry
points (`CholeskySolver`, `block_solve`, `schur_solve`), damping (`H_dd += ep + lm * H_dd`, :56-57),
"zero update when the factorisation fails" behaviour (:8-16) and implicit-function backward (:22-30).

Differences in construction, none in results:
  * the failure test is `torch.linalg.cholesky_ex`'s info flag, applied as a mask - no exception,
    no host synchronisation per solve;
  * the Schur complement is contracted block-wise over (pose, keyframe, pixel) with einsum instead
    of flattening E to a dense [6P, M*HW] matrix and transposing copies of it.
"""
import torch


class CholeskySolver(torch.autograd.Function):
    """x = H^-1 b for SPD H; x = 0 (and no gradient) where H is not SPD (chol.py:5-30)."""

    @staticmethod
    def forward(ctx, H, b):
        L, info = torch.linalg.cholesky_ex(H)
        ok = (info == 0).view(-1, 1, 1).to(b.dtype)
        L = torch.where(ok.bool(), L, torch.eye(H.shape[-1], dtype=H.dtype, device=H.device).expand_as(L))
        x = torch.cholesky_solve(b, L) * ok
        ctx.save_for_backward(L, x, ok)
        return x

    @staticmethod
    def backward(ctx, grad_x):
        L, x, ok = ctx.saved_tensors
        gb = torch.cholesky_solve(grad_x, L) * ok          # dL/db = H^-1 g
        gH = -torch.matmul(x, gb.transpose(-1, -2))         # dL/dH = -x (H^-1 g)^T
        return gH, gb


def _damp_diagonal(H, ep, lm):
    """H + (ep + lm*H) on the true diagonal only."""
    d = torch.diagonal(H, dim1=-2, dim2=-1)
    return H + torch.diag_embed(ep + lm * d)


def block_solve(H, b, ep=0.1, lm=0.0001):
    """Solve the pose-only system (chol.py:32-44).  H: [B,N,N,D,D], b: [B,N,D] -> [B,N,D].

    The reference damps with a [D,D] identity broadcast over *every* block (:35-36), i.e. the
    diagonals of the off-diagonal blocks receive `ep + lm*H` too; that is reproduced here.
    """
    B, N, _, D, _ = H.shape
    eye = torch.eye(D, dtype=H.dtype, device=H.device)
    H = H + (ep + lm * H) * eye
    Hd = H.permute(0, 1, 3, 2, 4).reshape(B, N * D, N * D)
    x = CholeskySolver.apply(Hd, b.reshape(B, N * D, 1))
    return x.reshape(B, N, D)


def schur_solve(H, E, C, v, w, ep=0.1, lm=0.0001, sless=False):
    """Eliminate the depth block, solve for poses, back-substitute (chol.py:47-73).

    H: [B,P,P,D,D] pose blocks, E: [B,P,M,D,HW] pose-depth blocks, C: [B,M,HW] depth diagonal,
    v: [B,P,D], w: [B,M,HW].  Returns dx [B,P,D] and dz [B,M,HW].
    """
    B, P, M, D, HW = E.shape
    Q = 1.0 / C
    EQ = E * Q[:, None, :, None, :]
    Hd = _damp_diagonal(H.permute(0, 1, 3, 2, 4).reshape(B, P * D, P * D), ep, lm)
    S = Hd - torch.einsum("bakdh,bckeh->badce", EQ, E).reshape(B, P * D, P * D)
    r = v - torch.einsum("bakdh,bkh->bad", EQ, w)
    dx = CholeskySolver.apply(S, r.reshape(B, P * D, 1)).reshape(B, P, D)
    if sless:
        return dx
    dz = Q * (w - torch.einsum("bakdh,bad->bkh", E, dx))
    return dx, dz
