"""CPU: the lookup oracle (restatement of correlation_kernels.cu:19-124) against an
independent formulation (bilinear grid_sample, zero padding) and its own adjoint."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O


def _case(seed, N=2, h1=5, w1=7, h2=6, w2=9, spread=4.0):
    g = np.random.default_rng(seed)
    vol = g.standard_normal((N, h1, w1, h2, w2)).astype(np.float32)
    base = np.stack(np.meshgrid(np.arange(w1), np.arange(h1)), 0).astype(np.float32)  # [2,h1,w1] x,y
    coords = base[None] * (w2 / w1) + g.uniform(-spread, spread, (N, 2, h1, w1)).astype(np.float32)
    return vol, coords.astype(np.float32)


def _grid_sample_lookup(vol, coords, r):
    """out[n,i,j,y,x] = bilinear(vol[n,y,x], (x0 - r + i, y0 - r + j)), zeros outside."""
    N, h1, w1, h2, w2 = vol.shape
    rd = 2 * r + 1
    v = torch.from_numpy(vol).reshape(N * h1 * w1, 1, h2, w2).double()
    c = torch.from_numpy(coords).permute(0, 2, 3, 1).reshape(N * h1 * w1, 1, 1, 2).double()
    d = torch.arange(-r, r + 1, dtype=torch.float64)
    dx = d.view(rd, 1).expand(rd, rd)  # first index i <-> x offset
    dy = d.view(1, rd).expand(rd, rd)  # second index j <-> y offset
    pts = c + torch.stack([dx, dy], -1).view(1, rd, rd, 2)
    gx = 2 * pts[..., 0] / (w2 - 1) - 1
    gy = 2 * pts[..., 1] / (h2 - 1) - 1
    out = F.grid_sample(v, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(N, h1, w1, rd, rd).permute(0, 3, 4, 1, 2).numpy()


@pytest.mark.parametrize("r", [1, 3])
@pytest.mark.parametrize("seed", [0, 1])
def test_forward_matches_bilinear_sampling(seed, r):
    vol, coords = _case(seed)
    got = O.corr_index_forward(vol, coords, r)
    ref = _grid_sample_lookup(vol, coords, r)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=0)


def test_forward_channel_order_is_x_major():
    # a single bright voxel: the output cell that sees it tells which index is the x offset
    N, h1, w1, h2, w2 = 1, 1, 1, 9, 9
    vol = np.zeros((N, h1, w1, h2, w2), np.float32)
    vol[0, 0, 0, 4, 6] = 1.0                       # y1 = 4, x1 = 6
    coords = np.array([4.0, 4.0], np.float32).reshape(1, 2, 1, 1)  # x0 = y0 = 4
    out = O.corr_index_forward(vol, coords, 3)[0, :, :, 0, 0]
    i, j = np.unravel_index(np.argmax(out), out.shape)
    assert (i, j) == (3 + 2, 3 + 0)                # first index follows x (correlation_kernels.cu:49,56-65)


def test_fp16_per_op_rounding_differs_from_fp32_then_round():
    vol, coords = _case(3)
    h = O.corr_index_forward(vol.astype(np.float16), coords, 3)
    f = O.corr_index_forward(vol.astype(np.float16).astype(np.float32), coords, 3, contract=False)
    assert h.dtype == np.float16
    np.testing.assert_allclose(h.astype(np.float32), f, atol=2e-2)
    assert np.any(h != f.astype(np.float16))      # the model is per-op rounding, not round-at-end


def test_out_of_bounds_and_empty():
    vol, coords = _case(4)
    coords[:] = 1e6
    assert not O.corr_index_forward(vol, coords, 3).any()
    coords[:] = -1e6
    assert not O.corr_index_forward(vol, coords, 3).any()
    e = O.corr_index_forward(np.zeros((0, 3, 3, 4, 4), np.float32), np.zeros((0, 2, 3, 3), np.float32), 3)
    assert e.shape == (0, 7, 7, 3, 3)


@pytest.mark.parametrize("r", [2, 3])
def test_backward_is_adjoint_of_forward(r):
    vol, coords = _case(5)
    g = np.random.default_rng(9).standard_normal((2, 2 * r + 1, 2 * r + 1, 5, 7))
    fwd = O.corr_index_forward(vol.astype(np.float64), coords, r)
    bwd = O.corr_index_backward(vol.shape, coords, g, r)
    np.testing.assert_allclose((fwd * g).sum(), (bwd * vol).sum(), rtol=1e-10)


def test_pyramid_lookup_concatenates_levels():
    g = np.random.default_rng(7)
    N, h1, w1 = 2, 4, 6
    pyr = [g.standard_normal((N, h1, w1, 8 >> l, 12 >> l)).astype(np.float32) for l in range(4)]
    coords = g.uniform(-2, 13, (N, h1, w1, 2)).astype(np.float32)
    out = O.corr_pyramid_lookup(pyr, coords, 3)
    assert out.shape == (N, 4 * 49, h1, w1)
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    for l in range(4):
        one = O.corr_index_forward(pyr[l], cf / 2 ** l, 3).reshape(N, 49, h1, w1)
        assert np.array_equal(out[:, 49 * l:49 * (l + 1)], one)


def test_half_conversion_is_ieee_rne():
    g = np.random.default_rng(11)
    x = np.concatenate([g.standard_normal(20000) * 10.0 ** g.integers(-9, 6, 20000),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.981e-8, np.inf, -np.inf]]).astype(np.float32)
    f = O.lib().oracle_f32_to_f16
    f.restype = __import__("ctypes").c_uint16
    f.argtypes = [__import__("ctypes").c_float]
    got = np.array([f(float(v)) for v in x], dtype=np.uint16)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16bit_lookup_equals_c10_half_arithmetic_op_by_op(dtype):
    """The oracle's software 16-bit arithmetic against c10::Half / c10::BFloat16 themselves: the reference kernel's statement
    sequence (correlation_kernels.cu:44-66) written with torch scalar-type tensors on the CPU, where every `*` and `+=` is
    ATen's half operator (convert to float, operate, round) - the same operators the CUDA kernel instantiates for scalar_t.
    Bit-exact, including the accumulation order into each output cell and the skipped out-of-range taps."""
    g = torch.Generator().manual_seed(4)
    N, h1, w1, h2, w2, r = 1, 3, 4, 6, 7, 3
    vol = torch.randn(N, h1, w1, h2, w2, generator=g).to(dtype)
    coords = torch.rand(N, 2, h1, w1, generator=g) * torch.tensor([w2 + 4.0, h2 + 4.0]).view(1, 2, 1, 1) - 2.0
    rd = 2 * r + 1
    out = torch.zeros(N, rd, rd, h1, w1, dtype=dtype)
    one = torch.tensor(1.0)
    for n in range(N):
        for y in range(h1):
            for x in range(w1):
                x0, y0 = coords[n, 0, y, x], coords[n, 1, y, x]
                dx, dy = x0 - torch.floor(x0), y0 - torch.floor(y0)
                w = {(0, 0): ((one - dx) * (one - dy)).to(dtype), (0, 1): ((one - dx) * dy).to(dtype),
                     (1, 0): (dx * (one - dy)).to(dtype), (1, 1): (dx * dy).to(dtype)}
                for i in range(rd + 1):
                    for j in range(rd + 1):
                        x1, y1 = int(torch.floor(x0)) - r + i, int(torch.floor(y0)) - r + j
                        if not (0 <= x1 < w2 and 0 <= y1 < h2):
                            continue
                        s = vol[n, y, x, y1, x1]
                        # the four statements of correlation_kernels.cu:56-65, in that order
                        if i > 0 and j > 0:
                            out[n, i - 1, j - 1, y, x] += s * w[(1, 1)]
                        if i > 0 and j < rd:
                            out[n, i - 1, j, y, x] += s * w[(1, 0)]
                        if i < rd and j > 0:
                            out[n, i, j - 1, y, x] += s * w[(0, 1)]
                        if i < rd and j < rd:
                            out[n, i, j, y, x] += s * w[(0, 0)]
    bf16 = dtype == torch.bfloat16
    if bf16:
        got = O.corr_index_forward(O.f32_to_bf16_bits(vol.float().numpy()), coords.numpy(), r, bf16=True)
        assert np.array_equal(got, O.f32_to_bf16_bits(out.float().numpy()))
    else:
        got = O.corr_index_forward(vol.numpy(), coords.numpy(), r)
        assert np.array_equal(got.view(np.uint16), out.numpy().view(np.uint16))
