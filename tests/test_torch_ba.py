"""The differentiable PyTorch BA (pvo_amd.geom.ba / chol / projective_ops) against fixtures produced
by the reference's own geom/ba.py (tests/golden/gen_golden.py): forward values, MoBA, gradients of
the custom Cholesky backward, failure behaviour, and - on the GPU - agreement with the HIP solver."""
import os

import numpy as np
import pytest
import torch

from pvo_amd.geom import projective_ops as pops
from pvo_amd.geom.ba import BA, MoBA
from pvo_amd.geom.chol import CholeskySolver, block_solve, schur_solve
from pvo_amd.geom.se3 import SE3

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, "ba_python_%s.npz" % name))
    t = {k: torch.from_numpy(z[k]) for k in z.files}
    P = t["poses"].shape[0]
    t["intr_all"] = t["intr"][None, None].repeat(1, P, 1)
    t["fixedp"] = int(z["fixedp"])
    return t


@pytest.mark.parametrize("name", ["a", "b"])
def test_projective_transform_matches_reference(name):
    t = _load(name)
    c, v = pops.projective_transform(SE3(t["poses"][None]), t["disps"][None], t["intr_all"], t["ii"], t["jj"])
    assert torch.allclose(c[0], t["reproj_coords"], atol=1e-5)
    assert torch.equal(v[0], t["reproj_valid"])
    c3, _ = pops.projective_transform(SE3(t["poses"][None]), t["disps"][None], t["intr_all"], t["ii"], t["jj"],
                                      return_depth=True)
    assert c3.shape[-1] == 3 and torch.allclose(c3[..., :2], c)


@pytest.mark.parametrize("name", ["a", "b"])
def test_jacobians_match_finite_differences(name):
    t = _load(name)
    poses, disps = SE3(t["poses"][None].double()), t["disps"][None].double()
    intr = t["intr_all"].double()
    ii, jj = t["ii"], t["jj"]
    c0, _, (Ji, Jj, Jz) = pops.projective_transform(poses, disps, intr, ii, jj, jacobian=True)
    eps = 1e-6
    for d in range(6):
        xi = torch.zeros(1, poses.shape[1], 6, dtype=torch.float64)
        xi[..., d] = eps
        c1, _ = pops.projective_transform(poses.retr(xi), disps, intr, ii, jj)
        num = (c1 - c0) / eps
        # perturbing every pose moves both ends of every edge
        assert torch.allclose(num, Ji[..., d] + Jj[..., d], atol=1e-4)
    c1, _ = pops.projective_transform(poses, disps + eps, intr, ii, jj)
    assert torch.allclose((c1 - c0) / eps, Jz[..., 0], atol=1e-4)


@pytest.mark.parametrize("name", ["a", "b"])
def test_ba_and_moba_match_reference(name):
    t = _load(name)
    Gs, disps = SE3(t["poses"][None].clone()), t["disps"][None].clone()
    for it in (1, 2):
        Gs, disps = BA(t["target"][None], t["weight"][None], t["eta"][None] - 1e-7, Gs, disps, t["intr_all"],
                       t["ii"], t["jj"], fixedp=t["fixedp"])
        assert torch.allclose(Gs.data[0], t["ba_poses_%d" % it], atol=1e-6)
        assert torch.allclose(disps[0], t["ba_disps_%d" % it], atol=1e-6)
    Gm = MoBA(t["target"][None], t["weight"][None], None, SE3(t["poses"][None].clone()), t["disps"][None],
              t["intr_all"], t["ii"], t["jj"], fixedp=t["fixedp"])
    assert torch.allclose(Gm.data[0], t["moba_poses_1"], atol=1e-6)
    # fixed poses do not move
    assert torch.equal(Gs.data[0, :t["fixedp"]], t["poses"][:t["fixedp"]])


@pytest.mark.parametrize("name", ["a", "b"])
def test_ba_gradients_match_reference(name):
    t = _load(name)
    leaves = {k: t[k][None].clone().requires_grad_(True) for k in ("target", "weight", "eta", "disps")}
    Gs, disps = SE3(t["poses"][None].clone()), leaves["disps"]
    for _ in range(2):
        Gs, disps = BA(leaves["target"], leaves["weight"], leaves["eta"] - 1e-7, Gs, disps, t["intr_all"],
                       t["ii"], t["jj"], fixedp=t["fixedp"])
    ((Gs.data[0] * t["grad_cp"]).sum() + (disps[0] * t["grad_cd"]).sum()).backward()
    for k, v in leaves.items():
        ref = t["grad_" + k]
        scale = ref.abs().max().item()
        assert (v.grad[0] - ref).abs().max().item() <= 2e-4 * scale + 1e-7, k


def test_cholesky_solver_failure_is_a_zero_update():
    H = torch.eye(6)[None].repeat(2, 1, 1)
    H[1, 0, 0] = -1.0                                   # second system is not SPD
    H.requires_grad_(True)
    b = torch.ones(2, 6, 1, requires_grad=True)
    x = CholeskySolver.apply(H, b)
    assert torch.equal(x[0], torch.ones(6, 1)) and torch.equal(x[1], torch.zeros(6, 1))
    x.sum().backward()
    assert torch.equal(b.grad[1], torch.zeros(6, 1)) and torch.equal(H.grad[1], torch.zeros(6, 6))
    assert torch.allclose(b.grad[0], torch.ones(6, 1))


def test_schur_solve_equals_dense_solve():
    g = torch.Generator().manual_seed(3)
    B, P, M, D, HW = 1, 3, 2, 6, 5
    J = torch.randn(B, P * D + M * HW, 60, generator=g, dtype=torch.float64)
    A = J @ J.transpose(1, 2)
    n = P * D
    H = A[:, :n, :n].reshape(B, P, D, P, D).permute(0, 1, 3, 2, 4)
    E = A[:, :n, n:].reshape(B, P, D, M, HW).permute(0, 1, 3, 2, 4)
    C = torch.rand(B, M, HW, generator=g, dtype=torch.float64) + 3000.0
    v = torch.randn(B, P, D, generator=g, dtype=torch.float64)
    w = torch.randn(B, M, HW, generator=g, dtype=torch.float64)
    dx, dz = schur_solve(H, E, C, v, w, ep=0.1, lm=1e-4)
    full = torch.zeros(B, n + M * HW, n + M * HW, dtype=torch.float64)
    Hd = A[:, :n, :n].clone()
    Hd = Hd + torch.diag_embed(0.1 + 1e-4 * torch.diagonal(Hd, dim1=-2, dim2=-1))
    full[:, :n, :n] = Hd
    full[:, :n, n:] = A[:, :n, n:]
    full[:, n:, :n] = A[:, :n, n:].transpose(1, 2)
    full[:, n:, n:] = torch.diag_embed(C.reshape(B, -1))
    sol = torch.linalg.solve(full, torch.cat([v.reshape(B, -1), w.reshape(B, -1)], 1)[..., None])[..., 0]
    assert torch.allclose(dx.reshape(B, -1), sol[:, :n], atol=1e-9)
    assert torch.allclose(dz.reshape(B, -1), sol[:, n:], atol=1e-9)
    assert block_solve(H, v).shape == (B, P, D)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b"])
def test_hip_ba_agrees_with_torch_ba(name):
    """Where the two reference BA paths coincide (all depths > 0.25, shared intrinsics) the HIP
    solver (fp64 solve, no +1e-7) and the torch BA take the same Gauss-Newton steps."""
    from pvo_amd import droid_backends as db
    t = _load(name)
    dev = torch.device("cuda:0")
    targets = t["target"].permute(0, 3, 1, 2).contiguous().to(dev)
    weights = t["weight"].permute(0, 3, 1, 2).contiguous().to(dev)
    P = t["poses"].shape[0]
    Gs, disps = SE3(t["poses"][None].clone()), t["disps"][None].clone()
    # (iterations, pose tolerance, depth tolerance).  The pose step of the first iteration is identical; the depth
    # step differs because the CUDA back-substitution skips window pose 0 (droid_kernels.cu:1084, reproduced by the
    # HIP kernel and pinned in test_ba_oracle.py), which then feeds into the second iteration's poses.
    for iters, ptol, dtol in ((1, 2e-5, 2e-2), (2, 1e-3, 2e-2)):
        Gs, disps = BA(t["target"][None], t["weight"][None], t["eta"][None] - 1e-7, Gs, disps, t["intr_all"],
                       t["ii"], t["jj"], fixedp=t["fixedp"])
        poses, d = t["poses"].clone().to(dev), t["disps"].clone().to(dev)
        db.ba(poses, d, t["intr"].to(dev), targets, weights, t["eta"].to(dev), t["ii"].to(dev), t["jj"].to(dev),
              t["fixedp"], P, iters, 1e-4, 0.1, False)
        assert torch.allclose(poses.cpu(), Gs.data[0], atol=ptol), iters
        assert torch.allclose(d.cpu(), disps[0], atol=dtol), iters


def _depth_only_case(H=12, W=20, seed=0):
    """BASELINE config 0 in miniature (S-1): two frames, two edges, both poses fixed -> a depth-only update"""
    g = torch.Generator().manual_seed(seed)
    intr = torch.tensor([W * 0.8, W * 0.8, W / 2.0, H / 2.0])
    xi = torch.tensor([0.06, 0.01, 0.03, 0.004, 0.012, -0.006])
    poses = torch.stack([SE3.exp(0 * xi).data, SE3.exp(xi).data])
    gt = 0.4 + 0.6 * torch.rand(2, H, W, generator=g)
    ii, jj = torch.tensor([0, 1]), torch.tensor([1, 0])
    c, _ = pops.projective_transform(SE3(poses[None]), gt[None], intr[None, None].repeat(1, 2, 1), ii, jj)
    weight = 200.0 * torch.rand(2, H, W, 2, generator=g)       # large confidences so that one step moves the depths visibly
    eta = 1e-3 + 1e-3 * torch.rand(2, H, W, generator=g)
    return dict(intr=intr, poses=poses, gt=gt, ii=ii, jj=jj, target=c[0], weight=weight, eta=eta, d0=torch.full((2, H, W), 0.7))


def test_depth_only_ba_with_all_poses_fixed_matches_oracle():
    from oracle import oracle as O
    s = _depth_only_case()
    intr_all = s["intr"][None, None].repeat(1, 2, 1)
    Gs, d = BA(s["target"][None], s["weight"][None], s["eta"][None] - 1e-7, SE3(s["poses"][None].clone()), s["d0"][None].clone(),
               intr_all, s["ii"], s["jj"], fixedp=2)
    assert torch.equal(Gs.data[0], s["poses"])                           # no free pose
    assert (d[0] - s["d0"]).abs().max() > 0.05                            # the depths did move ...
    assert (d[0] - s["gt"]).abs().mean() < 0.6 * (s["d0"] - s["gt"]).abs().mean()   # ... towards the truth
    r = O.ba(s["poses"].numpy().copy(), s["d0"].numpy().copy(), s["intr"].numpy(),
             s["target"].permute(0, 3, 1, 2).contiguous().numpy(), s["weight"].permute(0, 3, 1, 2).contiguous().numpy(),
             s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 2, 2, 1, 1e-4, 0.1)
    assert np.array_equal(r["poses"], s["poses"].numpy())
    assert np.abs(r["disps"] - d[0].numpy()).max() < 1e-4


@pytest.mark.gpu
def test_depth_only_ba_on_the_gpu_matches_oracle_and_torch():
    """P = t1 - t0 = 0: no pose block, no Schur complement, no solve - only dz = w / (C + eta)"""
    from oracle import oracle as O
    from pvo_amd import droid_backends as db
    s = _depth_only_case(H=47, W=156, seed=1)                            # the reference's CPU-runnable shape (config 0)
    dev = torch.device("cuda:0")
    poses, d = s["poses"].clone().to(dev), s["d0"].clone().to(dev)
    tg = s["target"].permute(0, 3, 1, 2).contiguous()
    wg = s["weight"].permute(0, 3, 1, 2).contiguous()
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    dx, dz = db.ba(poses, d, s["intr"].to(dev), tg.to(dev), wg.to(dev), s["eta"].to(dev), s["ii"].to(dev), s["jj"].to(dev),
                   2, 2, 1, 1e-4, 0.1, False, status=status)
    assert dx.shape == (0, 6) and dz.shape == (2, 47 * 156) and int(status[0]) == 0
    r = O.ba(s["poses"].numpy().copy(), s["d0"].numpy().copy(), s["intr"].numpy(), tg.numpy(), wg.numpy(), s["eta"].numpy(),
             s["ii"].numpy(), s["jj"].numpy(), 2, 2, 1, 1e-4, 0.1)
    assert torch.equal(poses.cpu(), s["poses"])
    assert np.abs(d.cpu().numpy() - r["disps"]).max() < 1e-4
    Gs, dt = BA(s["target"][None], s["weight"][None], s["eta"][None] - 1e-7, SE3(s["poses"][None].clone()), s["d0"][None].clone(),
                s["intr"][None, None].repeat(1, 2, 1), s["ii"], s["jj"], fixedp=2)
    assert (dt[0] - d.cpu()).abs().max() < 1e-4
