"""SE3 facade: the property tests of the reference's lietorch suite
(thirdparty/lietorch/lietorch/run_tests.py:16-54), fp64, atol 1e-8."""
import torch

from pvo_amd.geom.se3 import SE3


def _rand(n=64, sigma=1.0):
    torch.manual_seed(0)
    return SE3.Random(n, sigma=sigma, dtype=torch.float64)


def test_exp_log():
    torch.manual_seed(1)
    a = 0.2 * torch.randn(256, 6, dtype=torch.float64)
    assert torch.allclose(SE3.exp(a).log(), a, atol=1e-8)
    small = 1e-8 * torch.randn(16, 6, dtype=torch.float64)
    assert torch.allclose(SE3.exp(small).log(), small, atol=1e-12)


def test_inv():
    X = _rand()
    I = SE3.IdentityLike(X)
    assert torch.allclose((X * X.inv()).log(), I.log(), atol=1e-8)
    assert torch.allclose((X.inv() * X).data, I.data, atol=1e-8)


def test_adj():
    X = _rand()
    torch.manual_seed(2)
    a = torch.randn(64, 6, dtype=torch.float64)
    b = X.adj(a)
    Y1 = X * SE3.exp(a)
    Y2 = SE3.exp(b) * X
    assert torch.allclose((Y1 * Y2.inv()).log(), torch.zeros(64, 6, dtype=torch.float64), atol=1e-8)


def test_adjT_is_transpose_of_adj():
    X = _rand()
    torch.manual_seed(3)
    a = torch.randn(64, 6, dtype=torch.float64)
    b = torch.randn(64, 6, dtype=torch.float64)
    assert torch.allclose((X.adj(a) * b).sum(-1), (a * X.adjT(b)).sum(-1), atol=1e-10)


def test_act_matches_matrix():
    X = _rand()
    torch.manual_seed(4)
    p = torch.randn(64, 3, dtype=torch.float64)
    T = X.matrix()
    ph = torch.cat([p, torch.ones(64, 1, dtype=torch.float64)], -1)
    assert torch.allclose(X.act(p), (T @ ph[..., None])[..., :3, 0], atol=1e-10)
    q = torch.randn(64, 4, dtype=torch.float64)
    assert torch.allclose(X.act(q), (T @ q[..., None])[..., 0], atol=1e-10)


def test_retr_and_broadcast():
    X = _rand(8)
    a = torch.zeros(8, 6, dtype=torch.float64)
    assert torch.allclose(X.retr(a).data, X.data, atol=1e-12)
    p = torch.randn(8, 5, 7, 4, dtype=torch.float64)
    out = X[:, None, None] * p
    assert out.shape == p.shape
    assert torch.allclose(out[3, 2, 1], (X[3] * p[3, 2, 1]))


def test_matches_oracle_se3_helpers():
    """fp32 agreement with the restated CUDA helpers (droid_kernels.cu:58-107,856-874)."""
    import ctypes
    import numpy as np
    from oracle import oracle as O
    lib = O.lib()
    torch.manual_seed(5)
    Gi, Gj = SE3.Random(1, sigma=0.5).data[0], SE3.Random(1, sigma=0.5).data[0]
    out = np.zeros(7, np.float32)
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    gi, gj = Gi.numpy().copy(), Gj.numpy().copy()
    lib.oracle_relSE3(fp(gi), fp(gj), fp(out))
    want = (SE3(Gj) * SE3(Gi).inv()).data.numpy()
    assert np.allclose(out, want, atol=1e-6)
    a = np.random.default_rng(0).standard_normal(6).astype(np.float32)
    y = np.zeros(6, np.float32)
    lib.oracle_adjSE3(fp(gi), fp(a), fp(y))
    assert np.allclose(y, SE3(Gi).adjT(torch.from_numpy(a)).numpy(), atol=1e-5)
    xi = (0.05 * np.random.default_rng(1).standard_normal(6)).astype(np.float32)
    p1 = np.zeros(7, np.float32)
    lib.oracle_retrSE3(fp(xi), fp(gi), fp(p1), 0)
    assert np.allclose(p1, SE3(Gi).retr(torch.from_numpy(xi)).data.numpy(), atol=1e-6)
    # the xi[45] read (droid_kernels.cu:154) as "returned 0": translation differs visibly
    lib.oracle_retrSE3(fp(xi), fp(gi), fp(p1), 1)
    assert np.abs(p1[:3] - SE3(Gi).retr(torch.from_numpy(xi)).data.numpy()[:3]).max() > 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# Independent pins (not through this package's quaternion formulas): the matrix exponential of the 4x4 twist
# (torch.linalg.matrix_exp, fp64), scipy's quaternion -> rotation matrix, and explicit matrix algebra.  These are what
# keep the reference-BA fixtures (generated with this SE3 class standing in for lietorch) from being self-referential.
# ---------------------------------------------------------------------------------------------------------------------
def _hat(phi):
    z = torch.zeros_like(phi[..., 0])
    return torch.stack([torch.stack([z, -phi[..., 2], phi[..., 1]], -1),
                        torch.stack([phi[..., 2], z, -phi[..., 0]], -1),
                        torch.stack([-phi[..., 1], phi[..., 0], z], -1)], -2)


def _twist_matrix(xi):
    """lietorch's se(3) ordering: xi = (tau, phi)"""
    T = torch.zeros(xi.shape[:-1] + (4, 4), dtype=xi.dtype)
    T[..., :3, :3] = _hat(xi[..., 3:])
    T[..., :3, 3] = xi[..., :3]
    return T


def _check_against_matrices(device, dtype, atol):
    torch.manual_seed(11)
    xi = torch.cat([torch.randn(128, 3, dtype=torch.float64), 0.8 * torch.randn(128, 3, dtype=torch.float64)], -1)
    xi[:8, 3:] *= 1e-7                                                       # the small-angle branch
    M = torch.linalg.matrix_exp(_twist_matrix(xi))                            # ground truth, fp64
    X = SE3.exp(xi.to(device=device, dtype=dtype))
    assert torch.allclose(X.matrix().double().cpu(), M, atol=atol)
    # quaternion (xyzw) -> rotation: scipy's independent implementation
    from scipy.spatial.transform import Rotation
    q = X.data[..., 3:7].double().cpu().numpy()
    assert np.allclose(Rotation.from_quat(q).as_matrix(), M[:, :3, :3].numpy(), atol=atol)
    # group operations are matrix operations
    Y = SE3.exp(torch.flip(xi, [0]).to(device=device, dtype=dtype))
    MY = torch.flip(M, [0])
    assert torch.allclose((X * Y).matrix().double().cpu(), M @ MY, atol=10 * atol)
    assert torch.allclose(X.inv().matrix().double().cpu(), torch.linalg.inv(M), atol=10 * atol)
    p = torch.randn(128, 4, dtype=torch.float64)
    assert torch.allclose(X.act(p.to(device=device, dtype=dtype)).double().cpu(), (M @ p[..., None])[..., 0], atol=10 * atol)
    # Adjoint in the (tau, phi) ordering: [[R, hat(t) R], [0, R]]; adjT is its transpose (droid_kernels.cu:93-107 adjSE3)
    R, t = M[:, :3, :3], M[:, :3, 3]
    Ad = torch.zeros(128, 6, 6, dtype=torch.float64)
    Ad[:, :3, :3] = R; Ad[:, 3:, 3:] = R; Ad[:, :3, 3:] = _hat(t) @ R
    a = torch.randn(128, 6, dtype=torch.float64)
    assert torch.allclose(X.adj(a.to(device=device, dtype=dtype)).double().cpu(), (Ad @ a[..., None])[..., 0], atol=20 * atol)
    assert torch.allclose(X.adjT(a.to(device=device, dtype=dtype)).double().cpu(), (Ad.transpose(1, 2) @ a[..., None])[..., 0], atol=20 * atol)
    # log is the inverse of the matrix exponential; retr is left multiplication by Exp
    assert torch.allclose(torch.linalg.matrix_exp(_twist_matrix(X.log().double().cpu())), M, atol=20 * atol)
    assert torch.allclose(Y.retr(xi.to(device=device, dtype=dtype)).matrix().double().cpu(), M @ MY, atol=10 * atol)


import numpy as np
import pytest


def test_se3_against_matrix_exponential_and_scipy_fp64():
    _check_against_matrices("cpu", torch.float64, 1e-9)


def test_se3_against_matrix_exponential_and_scipy_fp32():
    _check_against_matrices("cpu", torch.float32, 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-9), (torch.float32, 2e-6)])
def test_se3_on_the_gpu_against_matrix_exponential(cuda, dtype, atol):
    _check_against_matrices(cuda, dtype, atol)


@pytest.mark.gpu
def test_se3_property_tests_on_the_gpu(cuda):
    """lietorch's run_tests.py:16-54 with device='cuda' (every test there takes a device argument)"""
    torch.manual_seed(0)
    X = SE3.Random(64, sigma=1.0, dtype=torch.float64, device=cuda)
    a = 0.2 * torch.randn(64, 6, dtype=torch.float64, device=cuda)
    assert torch.allclose(SE3.exp(a).log(), a, atol=1e-8)
    assert torch.allclose((X * X.inv()).log(), torch.zeros_like(a), atol=1e-8)
    Y1, Y2 = X * SE3.exp(a), SE3.exp(X.adj(a)) * X
    assert torch.allclose((Y1 * Y2.inv()).log(), torch.zeros_like(a), atol=1e-8)
    p = torch.randn(64, 4, dtype=torch.float64, device=cuda)
    assert torch.allclose(X.act(p), (X.matrix() @ p[..., None])[..., 0], atol=1e-10)


import contextlib


@contextlib.contextmanager
def torch_formulation():
    """the SE3 class on its PyTorch formulas (what CPU tensors always use): the reference of the HIP kernels, forward and backward"""
    from pvo_amd.geom import se3 as S
    old = S.FORCE_TORCH
    S.FORCE_TORCH = True
    try:
        yield
    finally:
        S.FORCE_TORCH = old


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,atol", [(torch.float64, 1e-12), (torch.float32, 2e-6)])
def test_native_se3_kernels_equal_the_torch_formulation(cuda, dtype, atol):
    """pvo_se3_unary / pvo_se3_binary (one fused kernel per operation, index broadcasting) against the torch formulation of
    the same class for every operation and for the broadcasting patterns the VO path uses (a pose per edge acting on H x W
    points, equal shapes, scalar pose)"""
    torch.manual_seed(3)
    B, N, H, W = 2, 5, 6, 7
    xi = torch.randn(B, N, 6, dtype=dtype, device=cuda) * 0.7
    xi[0, 0, 3:] *= 1e-8                                                       # small-angle branches
    p4 = torch.randn(B, N, H, W, 4, dtype=dtype, device=cuda)
    p3 = torch.randn(B, N, 3, dtype=dtype, device=cuda)
    a = torch.randn(B, N, H, W, 2, 6, dtype=dtype, device=cuda)

    def everything():
        X, Y = SE3.exp(xi), SE3.exp(torch.flip(xi, [1]))
        one = SE3(X.data[0, 0])                                                # a single pose against a batch
        return [X.data, X.log(), X.inv().data, (X * Y).data, X[:, :, None, None] * p4, X.act(p3), X[:, :, None, None, None].adjT(a),
                X.adj(a[:, :, 0, 0, 0]), (one * Y).data, Y.retr(xi).data]
    native = everything()
    with torch_formulation():
        ref = everything()
    for k, (g, r) in enumerate(zip(native, ref)):
        assert g.shape == r.shape and torch.allclose(g, r, atol=atol), k
    # a broadcast that is not an index broadcast falls back to the torch formulation
    q = torch.randn(1, N, H, 1, 4, dtype=dtype, device=cuda)
    out = SE3(native[0][:, :1, None, None]) * q
    assert out.shape == (B, N, H, 1, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
def test_native_se3_backward_kernels_equal_autograd_of_the_torch_formulation(cuda, dtype, rtol):
    """pvo_se3_unary_vjp / pvo_se3_binary_vjp (forward templates on dual numbers, one kernel per backward pass) against
    torch.autograd of the PyTorch formulation: every operation, both operands, index broadcasting (the gradient of a pose
    that acts on H x W points is the sum over them), the small-angle branches, and a composite expression of the kind the
    differentiable BA builds (Gj * Gi^-1 acting on points, retraction)."""
    torch.manual_seed(11)
    B, N, H, W = 2, 4, 5, 6
    leaves = dict(
        xi=torch.randn(B, N, 6, dtype=dtype, device=cuda) * 0.6,
        yi=torch.randn(B, N, 6, dtype=dtype, device=cuda) * 0.6,
        p4=torch.randn(B, N, H, W, 4, dtype=dtype, device=cuda),
        p3=torch.randn(B, N, 3, dtype=dtype, device=cuda),
        a=torch.randn(B, N, H, W, 2, 6, dtype=dtype, device=cuda),
        d=torch.randn(B, N, 6, dtype=dtype, device=cuda) * 0.1)
    leaves["xi"][0, 0, 3:] *= 1e-9                                             # (exp / log on their series branches)

    def run():
        L = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
        X, Y = SE3.exp(L["xi"]), SE3.exp(L["yi"])
        outs = [X.log(), X.inv().data, (X * Y).data, X[:, :, None, None] * L["p4"], X.act(L["p3"]),
                X[:, :, None, None, None].adjT(L["a"]), Y.adj(L["a"][:, :, 0, 0, 0]),
                ((Y * X.inv())[:, :, None, None] * L["p4"])[..., :3] / (1.0 + L["p4"][..., 3:] ** 2),      # relative pose acting on points
                Y.retr(L["d"]).data, (SE3(X.data[0, 0]) * Y).data]
        torch.manual_seed(5)
        loss = sum((o * torch.randn(o.shape, dtype=dtype, device=cuda)).sum() for o in outs)
        grads = torch.autograd.grad(loss, list(L.values()))
        return [o.detach() for o in outs], dict(zip(L.keys(), grads))
    outs_n, g_n = run()
    with torch_formulation():
        outs_t, g_t = run()
    for k, (x, y) in enumerate(zip(outs_n, outs_t)):
        assert torch.allclose(x, y, atol=1e-12 if dtype == torch.float64 else 2e-6), k
    for k in g_n:
        scale = g_t[k].abs().max().item()
        assert scale > 0 and (g_n[k] - g_t[k]).abs().max().item() <= rtol * scale, (k, (g_n[k] - g_t[k]).abs().max().item(), scale)


@pytest.mark.gpu
def test_native_se3_autograd_functions_are_used_when_a_gradient_is_recorded(cuda):
    x = (0.3 * torch.randn(7, 6, device=cuda)).requires_grad_(True)
    X = SE3.exp(x)
    assert X.data.requires_grad and "Unary" in type(X.data.grad_fn).__name__
    Y = X * X.inv()
    assert "Binary" in type(Y.data.grad_fn).__name__
    Y.data.sum().backward()
    assert torch.isfinite(x.grad).all()
