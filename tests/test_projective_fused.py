"""pvo_proj_transform / pvo_proj_transform_vjp (one fused HIP kernel per direction; pvo_amd/csrc/se3_ops.hip) against the PyTorch
formulation of projective_ops.projective_transform (reference: VO_Module/droid_slam/geom/projective_ops.py:106-130) - the same
function with PVO_SE3_TORCH semantics - forward and through torch.autograd.  fp64: to rounding; fp32: to fp32 rounding."""
import pytest
import torch

from pvo_amd.geom import projective_ops as pops
from pvo_amd.geom.se3 import SE3
from test_se3 import torch_formulation

pytestmark = pytest.mark.gpu


def _case(cuda, dtype, B=2, P=5, H=7, W=9):
    g = torch.Generator().manual_seed(3)
    xi = torch.randn(B, P, 6, generator=g, dtype=torch.float64) * torch.tensor([0.3, 0.3, 0.3, 0.15, 0.15, 0.15], dtype=torch.float64)
    xi[0, 1, :3] = torch.tensor([0.0, 0.0, -1.2])         # frame 1 of batch 0 far behind: points with Z < 0.1 and Z < 0.2 from it
    depths = torch.rand(B, P, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.05
    intr = torch.tensor([W * 0.8, W * 0.7, W / 2.0 - 0.3, H / 2.0 + 0.2], dtype=torch.float64).repeat(B, P, 1)
    intr[:, 2] *= 1.1                                      # (not every frame shares its intrinsics)
    ii = torch.tensor([0, 1, 1, 2, 3, 4, 2, 0])
    jj = torch.tensor([1, 0, 2, 1, 4, 3, 2, 3])           # includes an edge i == j
    to = lambda t: t.to(dtype).to(cuda)
    return to(xi), to(depths), to(intr), ii.to(cuda), jj.to(cuda)


def _run(xi, depths, intr, ii, jj, jacobian, return_depth, seed=7):
    xi, depths = xi.clone().requires_grad_(True), depths.clone().requires_grad_(True)
    poses = SE3.exp(xi)
    out = pops.projective_transform(poses, depths, intr, ii, jj, jacobian=jacobian, return_depth=return_depth)
    flat = [out[0], out[1]] + (list(out[2]) if jacobian else [])
    g = torch.Generator(device="cpu").manual_seed(seed)
    loss = sum((o * torch.randn(o.shape, generator=g, dtype=torch.float64).to(o)).sum() for k, o in enumerate(flat) if k != 1)
    gx, gd = torch.autograd.grad(loss, [xi, depths])
    return [o.detach() for o in flat], gx, gd


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
@pytest.mark.parametrize("jacobian,return_depth", [(True, False), (False, True), (False, False)])
def test_fused_projective_transform_equals_the_torch_formulation(cuda, dtype, tol, jacobian, return_depth):
    case = _case(cuda, dtype)
    got, gx, gd = _run(*case, jacobian, return_depth)
    with torch_formulation():
        want, wx, wd = _run(*case, jacobian, return_depth)
    assert len(got) == len(want) == (5 if jacobian else 2)
    for a, b in zip(got, want):
        assert a.shape == b.shape and a.dtype == b.dtype
    assert torch.equal(got[1], want[1]) and 0 < got[1].mean() < 1           # validity: both sides of Z > 0.2 occur
    assert got[0].shape[-1] == (3 if return_depth else 2)
    for k, (a, b) in enumerate(zip(got, want)):
        assert (a - b).abs().max() <= tol * max(1.0, b.abs().max().item()), k
    assert (gx - wx).abs().max() <= 10 * tol * max(1.0, wx.abs().max().item())
    assert (gd - wd).abs().max() <= 10 * tol * max(1.0, wd.abs().max().item())
    assert wx.abs().max() > 1e-3 and wd.abs().max() > 1e-3


def test_the_differentiable_ba_runs_on_the_fused_kernels_and_matches(cuda):
    """geom/ba.py's BA step (what DroidNet.forward differentiates through) with the fused projective_transform against the same
    step on the PyTorch formulation: updated poses / depths and the gradients of a loss on them with respect to the flow
    targets and weights"""
    from pvo_amd.geom.ba import BA
    xi, depths, intr, ii, jj = _case(cuda, torch.float64, B=1, P=5, H=6, W=8)
    keep = ii != jj
    ii, jj = ii[keep], jj[keep]
    N, H, W = ii.shape[0], 6, 8
    g = torch.Generator().manual_seed(1)
    tw = (torch.rand(1, N, H, W, 2, generator=g, dtype=torch.float64) * 0.5 + 0.25).to(cuda)
    eta = torch.full((1, int(torch.unique(ii).numel()), H, W), 1e-3, dtype=torch.float64, device=cuda)

    def run():
        poses = SE3.exp(xi * 0.3)
        coords, _ = pops.projective_transform(poses, depths, intr, ii, jj)
        target = (coords.detach() + 0.5).requires_grad_(True)
        weight = tw.clone().requires_grad_(True)
        p2, d2 = BA(target, weight, eta, poses, depths, intr, ii, jj, fixedp=1)
        loss = (p2.data * torch.arange(7, device=cuda, dtype=torch.float64)).sum() + (d2 ** 2).sum()
        gt, gw = torch.autograd.grad(loss, [target, weight])
        return p2.data.detach(), d2.detach(), gt, gw

    got = run()
    with torch_formulation():
        want = run()
    for a, b in zip(got, want):
        assert (a - b).abs().max() <= 1e-9 * max(1.0, b.abs().max().item())
    assert want[2].abs().max() > 1e-6


def test_frame_indices_outside_the_buffer_are_contained(cuda):
    """the fused kernels index poses / depths / intrinsics with ii, jj.  A negative index counts from the end, as in
    the torch formulation (which wraps it); an index still outside [0, P) - an IndexError there - gives that edge NaN coordinates,
    validity 0 and no gradient, instead of reads and atomics beside the buffers; the other edges are untouched."""
    from pvo_amd import droid_backends as db
    xi, depths, intr, ii, jj = _case(cuda, torch.float64)
    poses = SE3.exp(xi).data.contiguous()
    P = depths.shape[1]
    ref, ref_v = db.proj_transform(poses, depths, intr, ii, jj)
    neg_i = torch.where(ii == P - 1, torch.full_like(ii, -1), ii)                  # frame P - 1 spelled -1
    got, got_v = db.proj_transform(poses, depths, intr, neg_i, jj)
    assert torch.equal(got, ref) and torch.equal(got_v, ref_v)
    bad = jj.clone(); bad[2] = P + 3; bad[5] = -P - 1
    got, got_v, (Ji, Jj, Jz) = db.proj_transform(poses, depths, intr, ii, bad, jacobian=True)
    for n in range(ii.shape[0]):
        if n in (2, 5):
            assert torch.isnan(got[:, n]).all() and not got_v[:, n].any() and torch.isnan(Ji[:, n]).all()
        else:
            assert torch.equal(got[:, n], ref[:, n]) and torch.equal(got_v[:, n], ref_v[:, n])
    g = torch.ones_like(ref)
    gp, gd = db.proj_transform_vjp(poses, depths, intr, ii, bad, g, None, None, None)
    keep = torch.tensor([n for n in range(ii.shape[0]) if n not in (2, 5)], device=ii.device)
    gp2, gd2 = db.proj_transform_vjp(poses, depths, intr, ii[keep].contiguous(), jj[keep].contiguous(), g[:, keep].contiguous(), None, None, None)
    assert torch.isfinite(gp).all() and torch.allclose(gp, gp2, rtol=1e-12, atol=1e-12) and torch.allclose(gd, gd2, rtol=1e-12, atol=1e-12)
