"""Differentiable reprojection i -> j with closed-form Jacobians (PyTorch, any device).

Host-side mirror of the reference's ``geom/projective_ops.py`` (VO_Module/droid_slam/geom/
projective_ops.py:9-130): same function names, argument order, output shapes and thresholds
(``MIN_DEPTH = 0.2`` :6; ``Z < 0.1 -> 1`` :48; ``valid = Z1 > 0.2 & Z0 > 0.2`` :113).  This is the
training / autograd path (row 7 and 14 of SURVEY.md section 8a); inference goes through the HIP
kernels (`pvo_reproject`, `pvo_ba`).

Unlike the reference, no 4x6 point Jacobian or 2x4 projection Jacobian is materialised and
multiplied: the 2x6 product is written out in closed form (the same expressions the HIP
assemble kernel uses, pvo_amd/csrc/ba.hip), and the poses broadcast instead of being repeated
per pixel.
"""
import torch

from .se3 import SE3

MIN_DEPTH = 0.2


def extract_intrinsics(intrinsics):
    """[..., 4] -> four [..., 1, 1] tensors fx, fy, cx, cy (projective_ops.py:9-10)."""
    k = intrinsics[..., None, None, :]
    return k[..., 0], k[..., 1], k[..., 2], k[..., 3]


def coords_grid(ht, wd, **kwargs):
    """[ht, wd, 2] grid of (x, y) pixel coordinates (projective_ops.py:13-18)."""
    ys = torch.arange(ht, **kwargs).float()
    xs = torch.arange(wd, **kwargs).float()
    return torch.stack([xs[None, :].expand(ht, wd), ys[:, None].expand(ht, wd)], dim=-1)


def iproj(disps, intrinsics, jacobian=False):
    """Back-project to homogeneous points (X, Y, 1, d) (projective_ops.py:21-41)."""
    ht, wd = disps.shape[-2:]
    fx, fy, cx, cy = extract_intrinsics(intrinsics)
    grid = coords_grid(ht, wd, device=disps.device)
    X = ((grid[..., 0] - cx) / fx).expand_as(disps)
    Y = ((grid[..., 1] - cy) / fy).expand_as(disps)
    pts = torch.stack([X, Y, torch.ones_like(disps), disps], dim=-1)
    if not jacobian:
        return pts, None
    J = torch.zeros_like(pts)
    J[..., 3] = 1.0
    return pts, J


def _safe_inv_depth(Z):
    return 1.0 / torch.where(Z < 0.5 * MIN_DEPTH, torch.ones_like(Z), Z)


def proj(Xs, intrinsics, jacobian=False, return_depth=False):
    """Pinhole projection of homogeneous points (projective_ops.py:44-73)."""
    fx, fy, cx, cy = extract_intrinsics(intrinsics)
    X, Y, Z, D = Xs.unbind(dim=-1)
    d = _safe_inv_depth(Z)
    parts = [fx * (X * d) + cx, fy * (Y * d) + cy]
    if return_depth:
        parts.append(D * d)
    coords = torch.stack(parts, dim=-1)
    if not jacobian:
        return coords, None
    o = torch.zeros_like(d)
    J = torch.stack([torch.stack([fx * d, o, -fx * X * d * d, o], -1),
                     torch.stack([o, fy * d, -fy * Y * d * d, o], -1)], dim=-2)
    return coords, J


def actp(Gij, X0, jacobian=False):
    """Rigid action on a point cloud, optional 4x6 Jacobian (projective_ops.py:76-103)."""
    X1 = Gij[:, :, None, None] * X0
    if not jacobian:
        return X1, None
    X, Y, Z, d = X1.unbind(dim=-1)
    o = torch.zeros_like(d)
    rows = [torch.stack([d, o, o, o, Z, -Y], -1), torch.stack([o, d, o, -Z, o, X], -1),
            torch.stack([o, o, d, Y, -X, o], -1), torch.stack([o, o, o, o, o, o], -1)]
    return X1, torch.stack(rows, dim=-2)


class _ProjTransform(torch.autograd.Function):
    """projective_transform as one HIP kernel per direction (pvo_proj_transform / _vjp, pvo_amd/csrc/se3_ops.hip): the PyTorch
    formulation below is ~90 element-wise operators forward and as many backward, and a training step calls it ~75 times"""

    @staticmethod
    def forward(ctx, pdata, depths, intrinsics, ii, jj, jacobian, return_depth):
        from .. import droid_backends as db
        pdata, depths, intrinsics = pdata.contiguous(), depths.contiguous(), intrinsics.contiguous()
        ctx.save_for_backward(pdata, depths, intrinsics, ii, jj)
        ctx.jacobian = jacobian
        out = db.proj_transform(pdata, depths, intrinsics, ii, jj, jacobian, return_depth)
        valid = out[1].float()                              # (`.float()` in the reference, :113: fp32 whatever the inputs are)
        ctx.mark_non_differentiable(valid)
        return (out[0], valid) + tuple(out[2]) if jacobian else (out[0], valid)

    @staticmethod
    def backward(ctx, g_x1, g_valid, g_Ji=None, g_Jj=None, g_Jz=None):
        from .. import droid_backends as db
        pdata, depths, intrinsics, ii, jj = ctx.saved_tensors
        gp, gd = db.proj_transform_vjp(pdata, depths, intrinsics, ii, jj, g_x1, g_Ji, g_Jj, g_Jz)
        return gp, gd, None, None, None, None, None


def _fused(poses, depths, intrinsics):
    from . import se3
    d = poses.data
    return (not se3.FORCE_TORCH and depths.is_cuda and d.is_cuda and intrinsics.is_cuda and depths.dim() == 4 and d.dim() == 3
            and depths.dtype in (torch.float32, torch.float64) and d.dtype == depths.dtype and intrinsics.dtype == depths.dtype
            and not intrinsics.requires_grad)


def projective_transform(poses, depths, intrinsics, ii, jj, jacobian=False, return_depth=False):
    """Map the pixels of frames ``ii`` into frames ``jj`` (projective_ops.py:106-130).

    poses: SE3 [B, P] (world-to-camera); depths: [B, P, H, W] inverse depth; intrinsics [B, P, 4].
    Returns coords [B, N, H, W, 2(3)], valid [B, N, H, W, 1] and, with ``jacobian``, the tuple
    (Ji [B,N,H,W,2,6], Jj [B,N,H,W,2,6], Jz [B,N,H,W,2,1]).
    Device tensors (fp32 / fp64) take one fused kernel per direction; the formulation below is its reference, what CPU tensors
    use, and what `pvo_amd.config.debug_config("se3_torch", True)` selects everywhere.
    """
    if _fused(poses, depths, intrinsics):
        dev = depths.device
        ii_d = torch.as_tensor(ii, dtype=torch.long, device=dev).contiguous()
        jj_d = torch.as_tensor(jj, dtype=torch.long, device=dev).contiguous()
        out = _ProjTransform.apply(poses.data, depths, intrinsics, ii_d, jj_d, bool(jacobian), bool(return_depth))
        return (out[0], out[1], (out[2], out[3], out[4])) if jacobian else (out[0], out[1])
    X0, _ = iproj(depths[:, ii], intrinsics[:, ii])
    Gij = poses[:, jj] * poses[:, ii].inv()
    X1 = Gij[:, :, None, None] * X0
    X, Y, Z, W = X1.unbind(dim=-1)

    fx, fy, cx, cy = extract_intrinsics(intrinsics[:, jj])
    d = _safe_inv_depth(Z)
    parts = [fx * (X * d) + cx, fy * (Y * d) + cy]
    if return_depth:
        parts.append(W * d)
    x1 = torch.stack(parts, dim=-1)
    valid = ((Z > MIN_DEPTH) & (X0[..., 2] > MIN_DEPTH)).float().unsqueeze(-1)
    if not jacobian:
        return x1, valid

    # d(coords)/d(xi_j): rows of Jp @ Ja written out (Ja uses the true Z, Jp the guarded 1/Z)
    fxd, fyd = fx * d, fy * d
    gx, gy = -fx * X * d * d, -fy * Y * d * d       # d(x)/dZ, d(y)/dZ
    o = torch.zeros_like(d)
    Jj = torch.stack([
        torch.stack([fxd * W, o, gx * W, gx * Y, fxd * Z - gx * X, -fxd * Y], -1),
        torch.stack([o, fyd * W, gy * W, -fyd * Z + gy * Y, -gy * X, fyd * X], -1)], dim=-2)
    # pose i enters through Gi^-1: dual adjoint of the relative pose, negated
    Ji = -Gij[:, :, None, None, None].adjT(Jj)
    # depth enters through the translation column of Gij
    t = Gij.data[..., None, None, :3]
    Jz = torch.stack([fxd * t[..., 0] + gx * t[..., 2], fyd * t[..., 1] + gy * t[..., 2]], dim=-1)
    return x1, valid, (Ji, Jj, Jz.unsqueeze(-1))


def projective_transform_unsup(poses, depths, intrinsics, ii, jj):
    """coords, the points' inverse depth in frame j, valid (projective_ops.py:133-163): what the unsupervised occlusion
    masks compare with frame j's own depth map"""
    x1, valid = projective_transform(poses, depths, intrinsics, ii, jj, jacobian=False, return_depth=True)
    return x1[..., 0:2], x1[..., 2:3], valid


def induced_flow(poses, disps, intrinsics, ii, jj):
    """Optical flow induced by camera motion (projective_ops.py, `induced_flow`)."""
    ht, wd = disps.shape[2:]
    grid = coords_grid(ht, wd, device=disps.device)
    coords1, valid = projective_transform(poses, disps, intrinsics, ii, jj, False)
    return coords1[..., :2] - grid, valid
