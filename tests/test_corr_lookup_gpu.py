"""GPU parity: the HIP lookup (through droid_backends -> C ABI) against the CPU oracle.
Bit-exact for every dtype (integer index math + the reference's rounding model)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

DT = {"f32": (np.float32, torch.float32), "f16": (np.float16, torch.float16), "f64": (np.float64, torch.float64)}


def _inputs(seed, N, h1, w1, h2, w2, spread=5.0):
    g = np.random.default_rng(seed)
    vol = g.standard_normal((N, h1, w1, h2, w2)).astype(np.float32)
    base = np.stack(np.meshgrid(np.arange(w1), np.arange(h1)), 0).astype(np.float32)
    coords = base[None] * (w2 / max(w1, 1)) + g.uniform(-spread, spread, (N, 2, h1, w1)).astype(np.float32)
    # a few exact-integer and far-out-of-range samples
    flat = coords.reshape(-1)
    flat[::17] = np.round(flat[::17])
    flat[5::101] = 1e7
    flat[7::103] = -3.5
    return vol, coords.astype(np.float32)


def _bits(a):
    return a.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


@pytest.mark.parametrize("dt", ["f32", "f16", "f64"])
@pytest.mark.parametrize("r", [3, 1])
@pytest.mark.parametrize("shape", [(2, 5, 7, 6, 9), (3, 8, 16, 8, 16), (1, 9, 13, 3, 5), (2, 4, 64, 12, 16)])
def test_corr_index_forward_bit_exact(cuda, dt, r, shape):
    from pvo_amd import droid_backends as db
    npd, td = DT[dt]
    vol, coords = _inputs(sum(shape) + r, *shape)
    vol = vol.astype(npd)
    want = O.corr_index_forward(vol, coords, r)
    got, = db.corr_index_forward(torch.from_numpy(vol).to(cuda), torch.from_numpy(coords).to(cuda), r)
    assert got.dtype == td and tuple(got.shape) == want.shape
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("r", [3, 2])
def test_corr_index_forward_bf16_bit_exact(cuda, r):
    from pvo_amd import droid_backends as db
    vol, coords = _inputs(21, 2, 6, 10, 7, 11)
    vb = torch.from_numpy(vol).to(torch.bfloat16)
    bits = vb.view(torch.int16).numpy().view(np.uint16)
    want = O.corr_index_forward(bits, coords, r, bf16=True)
    got, = db.corr_index_forward(vb.to(cuda), torch.from_numpy(coords).to(cuda), r)
    assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16), want)


def test_unaligned_16bit_view_uses_generic_path(cuda):
    from pvo_amd import droid_backends as db
    vol, coords = _inputs(5, 2, 5, 7, 6, 9)
    vol = vol.astype(np.float16)
    buf = torch.zeros(vol.size + 1, dtype=torch.float16, device=cuda)
    buf[1:] = torch.from_numpy(vol).to(cuda).reshape(-1)
    v = buf[1:].view(*vol.shape)          # 2-byte aligned only
    assert v.data_ptr() % 4 == 2 and v.is_contiguous()
    got, = db.corr_index_forward(v, torch.from_numpy(coords).to(cuda), 3)
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(O.corr_index_forward(vol, coords, 3)))


def test_edge_cases(cuda):
    from pvo_amd import droid_backends as db
    # empty graph
    got, = db.corr_index_forward(torch.zeros(0, 3, 3, 4, 4, device=cuda), torch.zeros(0, 2, 3, 3, device=cuda), 3)
    assert tuple(got.shape) == (0, 7, 7, 3, 3)
    # everything out of bounds -> exact zeros; tiny target plane (smaller than the window)
    vol = torch.randn(1, 4, 4, 2, 3, device=cuda, dtype=torch.float16)
    c = torch.full((1, 2, 4, 4), 500.0, device=cuda)
    got, = db.corr_index_forward(vol, c, 3)
    assert not got.any()
    c = torch.rand(1, 2, 4, 4, device=cuda) * 3
    got, = db.corr_index_forward(vol, c, 3)
    want = O.corr_index_forward(vol.cpu().numpy(), c.cpu().numpy(), 3)
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))
    # contiguity error behaviour (droid.cpp:83)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        db.corr_index_forward(vol.transpose(1, 2), c, 3)
    # no CPU fallback
    with pytest.raises(RuntimeError):
        db.corr_index_forward(vol.cpu(), c.cpu(), 3)


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_pyramid_lookup_config_shapes(cuda, dt):
    """Fused 4-level lookup == oracle of CorrBlock.__call__; BASELINE config-1 style odd map (7x13 of 47x156 scaled down)."""
    from pvo_amd import droid_backends as db
    npd, td = DT[dt]
    g = np.random.default_rng(3)
    N, h1, w1 = 3, 11, 19
    pyr = [g.standard_normal((N, h1, w1, h1 >> l, w1 >> l)).astype(npd) for l in range(4)]
    coords = (np.stack(np.meshgrid(np.arange(w1), np.arange(h1)), -1)[None].astype(np.float32)
              + g.uniform(-6, 6, (N, h1, w1, 2)).astype(np.float32))
    want = O.corr_pyramid_lookup(pyr, coords, 3)
    got = db.corr_pyramid_lookup([torch.from_numpy(p).to(cuda) for p in pyr], torch.from_numpy(coords).to(cuda), 3)
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))


def test_pyramid_lookup_full_size_sb(cuda):
    """S-B (BASELINE configs[1]): 48x64 maps, fp16; ALL 36 edges against the oracle bit-exactly (round 2 compared 4),
    and a size-independent property (a constant volume returns the in-bounds bilinear mass)."""
    from pvo_amd import droid_backends as db
    g = torch.Generator().manual_seed(0)
    N, H, W = 36, 48, 64
    pyr = [torch.randn(N, H, W, H >> l, W >> l, generator=g).half() for l in range(4)]
    base = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float()
    coords = base[None] + torch.randn(N, H, W, 2, generator=g) * 4
    got = db.corr_pyramid_lookup([p.to(cuda) for p in pyr], coords.to(cuda), 3).cpu()
    assert tuple(got.shape) == (N, 196, H, W)
    want = O.corr_pyramid_lookup([p.numpy() for p in pyr], coords.numpy(), 3)
    assert np.array_equal(_bits(got.numpy()), _bits(want))
    ones = [torch.ones_like(p).to(cuda) for p in pyr]
    o = db.corr_pyramid_lookup(ones, coords.to(cuda), 3).cpu().float()
    inside = (coords[..., 0] > 4) & (coords[..., 0] < W - 5) & (coords[..., 1] > 4) & (coords[..., 1] < H - 5)
    lvl0 = o[:, :49].permute(0, 2, 3, 1)[inside]
    assert torch.allclose(lvl0, torch.ones_like(lvl0), atol=2e-3)


@pytest.mark.parametrize("dt", ["f32", "f16", "f64"])
@pytest.mark.parametrize("r", [3, 2])
def test_corr_index_backward_bit_exact(cuda, dt, r):
    from pvo_amd import droid_backends as db
    npd, td = DT[dt]
    vol, coords = _inputs(77, 2, 6, 9, 7, 10)
    g = np.random.default_rng(8).standard_normal((2, 2 * r + 1, 2 * r + 1, 6, 9)).astype(npd)
    want = O.corr_index_backward(vol.shape, coords, g, r)
    got, = db.corr_index_backward(torch.from_numpy(vol.astype(npd)).to(cuda), torch.from_numpy(coords).to(cuda),
                                  torch.from_numpy(g).to(cuda), r)
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_pyramid_lookup_full_size_s1(cuda, dt):
    """S-1 (BASELINE configs[0], evaluation_scripts/test_vo2.py: VKITTI2 376x1248 / 8 = 47x156 maps, 2 frames, 2 edges):
    the 4-level lookup in the reference's own volume layout at FULL map size against the oracle, bit-exact, fp32 (what
    DroidNet.forward correlates in) and fp16."""
    from pvo_amd import droid_backends as db
    npd, td = DT[dt]
    g = np.random.default_rng(5)
    N, h1, w1 = 2, 47, 156
    pyr = [g.standard_normal((N, h1, w1, h1 >> l, w1 >> l), dtype=np.float32).astype(npd) for l in range(4)]
    coords = (np.stack(np.meshgrid(np.arange(w1), np.arange(h1)), -1)[None].astype(np.float32)
              + g.uniform(-9, 9, (N, h1, w1, 2)).astype(np.float32))
    coords[0, 0, :4] = np.array([[-3.0, -3.0], [w1 + 2.5, h1 + 2.5], [0.0, 0.0], [w1 - 1.0, h1 - 1.0]], np.float32)
    want = O.corr_pyramid_lookup(pyr, coords, 3)
    got = db.corr_pyramid_lookup([torch.from_numpy(p).to(cuda) for p in pyr], torch.from_numpy(coords).to(cuda), 3)
    assert tuple(got.shape) == (N, 196, h1, w1)
    assert np.array_equal(_bits(got.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("dt,N,H,W", [("f16", 2, 48, 64), ("f32", 1, 48, 64), ("f16", 1, 30, 101), ("f32", 1, 47, 156)])
def test_corr_index_backward_bit_exact_at_full_map_size(cuda, dt, N, H, W):
    """corr_index_backward (correlation_kernels.cu:73-124) at the S-B / S-A / S-1 map sizes (round 2 checked 6x9 only): the
    dense volume gradient of level 0 and of level 2 (odd pooled sizes), bit for bit against the oracle."""
    from pvo_amd import droid_backends as db
    npd, td = DT[dt]
    g = np.random.default_rng(H * W)
    for l in (0, 2):
        h2, w2 = H >> l, W >> l
        coords = (np.stack(np.meshgrid(np.arange(W), np.arange(H)), 0)[None].astype(np.float32) / (1 << l)
                  + g.uniform(-5, 5, (N, 2, H, W)).astype(np.float32)).astype(np.float32)
        gr = g.standard_normal((N, 7, 7, H, W), dtype=np.float32).astype(npd)
        want = O.corr_index_backward((N, H, W, h2, w2), coords, gr, 3)
        vol = torch.empty(N, H, W, h2, w2, dtype=td, device=cuda)
        got, = db.corr_index_backward(vol, torch.from_numpy(coords).to(cuda), torch.from_numpy(gr).to(cuda), 3)
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(want)), l
