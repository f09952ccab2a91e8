"""Volume-free correlation (altcorr_kernel.cu:27-286).
CPU: the oracle against the volume path it must agree with (dot products == volume entries, same
bilinear lookup, same channel order) and against a finite-difference adjoint.
GPU: HIP forward/backward against the oracle (fp32: accumulation order differs -> 1e-4 relative)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O


def _case(seed, B=2, S=2, H1=5, W1=7, H2=6, W2=9, C=32):
    g = np.random.default_rng(seed)
    f1 = g.standard_normal((B, H1, W1, C)).astype(np.float32)
    f2 = g.standard_normal((B, H2, W2, C)).astype(np.float32)
    base = np.stack(np.meshgrid(np.arange(W1), np.arange(H1)), -1).astype(np.float32) * (W2 / W1)
    coords = base[None, None] + g.uniform(-4, 4, (B, S, H1, W1, 2)).astype(np.float32)
    return f1, f2, coords.astype(np.float32)


def test_oracle_altcorr_equals_lookup_of_the_explicit_volume():
    f1, f2, coords = _case(0)
    B, S = coords.shape[:2]
    got = O.altcorr_forward(f1, f2, coords, 3)
    vol = np.einsum("bhwc,byxc->bhwyx", f1.astype(np.float64), f2.astype(np.float64))      # [B,H1,W1,H2,W2]
    for s in range(S):
        c = np.ascontiguousarray(coords[:, s].transpose(0, 3, 1, 2))
        want = O.corr_index_forward(vol, c, 3)                                             # [B,7(x),7(y),H1,W1]
        # altcorr channel = iy + 7*ix: x offset major, the same order as the volume lookup
        assert np.allclose(got[:, s].reshape(B, 7, 7, *got.shape[-2:]), want, atol=1e-4)


def test_oracle_altcorr_backward_is_adjoint():
    f1, f2, coords = _case(1)
    g = np.random.default_rng(5).standard_normal((2, 2, 49, 5, 7)).astype(np.float32)
    g1, g2 = O.altcorr_backward(f1, f2, coords, g, 3)
    d1 = np.random.default_rng(6).standard_normal(f1.shape).astype(np.float32)
    d2 = np.random.default_rng(7).standard_normal(f2.shape).astype(np.float32)
    eps = 1e-2
    fd = ((O.altcorr_forward(f1 + eps * d1, f2 + eps * d2, coords, 3).astype(np.float64)
           - O.altcorr_forward(f1 - eps * d1, f2 - eps * d2, coords, 3)) * g).sum() / (2 * eps)
    an = (g1.astype(np.float64) * d1).sum() + (g2.astype(np.float64) * d2).sum()
    assert abs(fd - an) < 1e-3 * max(1.0, abs(an))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 2, 5, 7, 6, 9, 32, 3), (1, 3, 9, 13, 4, 6, 128, 3), (2, 1, 8, 70, 8, 70, 20, 3), (1, 2, 6, 6, 6, 6, 16, 2)])
def test_hip_altcorr_forward_backward_match_oracle(cuda, cfg):
    from pvo_amd import droid_backends as db
    B, S, H1, W1, H2, W2, C, r = cfg
    f1, f2, coords = _case(sum(cfg), B, S, H1, W1, H2, W2, C)
    coords.reshape(-1)[3::37] = 1e6                                   # some far-away samples
    t = lambda a: torch.from_numpy(a).to(cuda)
    got, = db.altcorr_forward(t(f1), t(f2), t(coords), r)
    want = O.altcorr_forward(f1, f2, coords, r)
    assert np.allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    rd = 2 * r + 1
    g = np.random.default_rng(2).standard_normal((B, S, rd * rd, H1, W1)).astype(np.float32)
    g1, g2, gc = db.altcorr_backward(t(f1), t(f2), t(coords), t(g), r)
    w1, w2 = O.altcorr_backward(f1, f2, coords, g, r)
    assert np.allclose(g1.cpu().numpy(), w1, rtol=1e-4, atol=1e-4)
    assert np.allclose(g2.cpu().numpy(), w2, rtol=1e-4, atol=2e-4)
    assert not gc.any() and tuple(gc.shape) == (B, S, H1, W1, 2)      # altcorr_kernel.cu:340


@pytest.mark.gpu
def test_altcorrblock_matches_corrblock_lookup(cuda):
    """AltCorrBlock (on-the-fly) and CorrBlock (precomputed volume) are two routes to the same 196
    channels; fp32 features so that neither side rounds to fp16."""
    from pvo_amd.modules.corr import AltCorrBlock, CorrBlock
    g = torch.Generator().manual_seed(0)
    N, C, H, W = 3, 64, 16, 24
    fmaps = torch.randn(1, N, C, H, W, generator=g).to(cuda)
    ii, jj = torch.tensor([0, 1, 2, 0], device=cuda), torch.tensor([1, 2, 0, 2], device=cuda)
    base = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float()
    coords = (base[None, None] + torch.randn(1, 4, H, W, 2, generator=g) * 3).to(cuda)
    with torch.no_grad():
        a = AltCorrBlock(fmaps)(coords, ii, jj)
        b = CorrBlock(fmaps[:, ii], fmaps[:, jj])(coords)
    assert a.shape == b.shape == (1, 4, 196, H, W)
    # the dot product is linear in fmap2, so pooling the features (AltCorr) and pooling the volume
    # (CorrBlock) give the same pyramid: all 4 x 49 channels agree up to fp32 summation order
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-3)
