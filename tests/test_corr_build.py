"""Correlation volume + pyramid: oracle vs the reference's CorrBlock fixture (CPU), and the
HIP build vs the oracle / the fixture / its own pooling identity (GPU).

Tolerances: the reference computes level 0 with torch.matmul (fp32 accumulation in an
unspecified order) and rounds to the storage type, so level 0 is compared to one unit in the
last place of the storage type; pooled levels are bit-exact functions of the ROUNDED level
below, which is checked exactly."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _ulp_close(a, b, dtype):
    a = a.astype(np.float64); b = b.astype(np.float64)
    eps = {np.float16: 2.0 ** -10, np.float32: 2.0 ** -23}[dtype]
    return np.all(np.abs(a - b) <= 1.01 * eps * np.maximum(np.abs(b), 2.0 ** -14) + 1e-7)


def test_oracle_matches_reference_corrblock_fixture():
    d = np.load(os.path.join(G, "corr_volume.npz"))
    lv = O.corr_build(d["fmap1"], d["fmap2"], 4)
    for l in range(4):
        ref = d["level%d_f32" % l]
        assert lv[l].shape == ref.shape
        assert np.allclose(lv[l], ref, rtol=2e-5, atol=2e-5)
    lh = O.corr_build(d["fmap1"].astype(np.float16), d["fmap2"].astype(np.float16), 4)
    assert _ulp_close(lh[0], d["level0_f16"], np.float16)
    for l in range(1, 4):
        assert np.abs(lh[l].astype(np.float32) - d["level%d_f16" % l].astype(np.float32)).max() < 4e-3


def _pool_exact(level, dtype):
    """2x2 mean of a stored level, fp32 sum in window order, one rounding (ATen avg_pool2d)."""
    v = level.astype(np.float32)
    h, w = v.shape[-2] // 2, v.shape[-1] // 2
    v = v[..., :2 * h, :2 * w]
    s = ((v[..., 0::2, 0::2] + v[..., 0::2, 1::2]) + v[..., 1::2, 0::2]) + v[..., 1::2, 1::2]
    return (s * np.float32(0.25)).astype(dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 128, 16, 16), (1, 64, 11, 19), (2, 32, 8, 40), (1, 128, 24, 33), (3, 16, 5, 7)])
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
def test_mfma_build_matches_oracle_and_pools_exactly(cuda, shape, tdt):
    from pvo_amd import droid_backends as db
    N, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    f1 = torch.randn(N, C, H, W, generator=g).to(tdt)
    f2 = torch.randn(N, C, H, W, generator=g).to(tdt)
    got = db.corr_build(f1.permute(0, 2, 3, 1).contiguous().to(cuda), f2.permute(0, 2, 3, 1).contiguous().to(cuda),
                        4, channels_last=True)
    assert [tuple(x.shape) for x in got] == [(N, H, W, H >> l, W >> l) for l in range(4)]
    ref = torch.einsum("ncp,ncq->npq", f1.double().flatten(2) / 4, f2.double().flatten(2) / 4).view(N, H, W, H, W)
    eps = 2.0 ** -10 if tdt == torch.float16 else 2.0 ** -7
    l0 = got[0].cpu().double()
    assert torch.all((l0 - ref).abs() <= 1.01 * eps * ref.abs().clamp(min=2.0 ** -14) + 1e-4)
    # pooled levels are exact functions of the stored level below
    for l in range(1, 4):
        below = got[l - 1].cpu()
        if tdt == torch.float16:
            want = _pool_exact(below.numpy(), np.float16)
            assert np.array_equal(got[l].cpu().numpy().view(np.uint16), want.view(np.uint16))
        else:
            v = below.float()
            h, w = v.shape[-2] // 2, v.shape[-1] // 2
            v = v[..., :2 * h, :2 * w]
            s = ((v[..., 0::2, 0::2] + v[..., 0::2, 1::2]) + v[..., 1::2, 0::2]) + v[..., 1::2, 1::2]
            assert torch.equal(got[l].cpu(), (s * 0.25).to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("tdt,npd", [(torch.float32, np.float32), (torch.float64, np.float64), (torch.float16, np.float16)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_generic_build_matches_oracle(cuda, tdt, npd, channels_last):
    from pvo_amd import droid_backends as db
    N, C, H, W = 2, 24, 9, 13          # C % 16 != 0 forces the generic path for fp16 too
    g = np.random.default_rng(4)
    f1 = g.standard_normal((N, C, H, W)).astype(npd)
    f2 = g.standard_normal((N, C, H, W)).astype(npd)
    want = O.corr_build(f1, f2, 4)
    t1, t2 = torch.from_numpy(f1).to(cuda), torch.from_numpy(f2).to(cuda)
    if channels_last:
        t1, t2 = t1.permute(0, 2, 3, 1).contiguous(), t2.permute(0, 2, 3, 1).contiguous()
    got = db.corr_build(t1, t2, 4, channels_last=channels_last)
    for l in range(4):
        a, b = got[l].cpu().numpy().astype(np.float64), want[l].astype(np.float64)
        tol = {np.float32: 1e-5, np.float64: 1e-5, np.float16: 2e-3}[npd]   # the oracle accumulates in fp32
        assert a.shape == b.shape and np.allclose(a, b, rtol=tol, atol=tol)


@pytest.mark.gpu
def test_build_matches_reference_fixture(cuda):
    from pvo_amd import droid_backends as db
    d = np.load(os.path.join(G, "corr_volume.npz"))
    f1 = torch.from_numpy(d["fmap1"]).half().permute(0, 2, 3, 1).contiguous().to(cuda)
    f2 = torch.from_numpy(d["fmap2"]).half().permute(0, 2, 3, 1).contiguous().to(cuda)
    got = db.corr_build(f1, f2, 4, channels_last=True)
    assert _ulp_close(got[0].cpu().numpy(), d["level0_f16"], np.float16)
    for l in range(1, 4):
        assert np.abs(got[l].cpu().float().numpy() - d["level%d_f16" % l].astype(np.float32)).max() < 4e-3
    f1 = torch.from_numpy(d["fmap1"]).to(cuda); f2 = torch.from_numpy(d["fmap2"]).to(cuda)
    got = db.corr_build(f1, f2, 4)
    for l in range(4):
        assert np.allclose(got[l].cpu().numpy(), d["level%d_f32" % l], rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_build_then_lookup_full_size_sb(cuda):
    """S-B shapes (48x64, C=128, fp16): volume of an edge from a frame to ITSELF peaks on the
    diagonal, so a lookup at the identity grid returns the self-correlation in the centre tap."""
    from pvo_amd import droid_backends as db
    g = torch.Generator().manual_seed(0)
    N, C, H, W = 4, 128, 48, 64
    f = torch.randn(N, H, W, C, generator=g).half().to(cuda)
    pyr = db.corr_build(f, f, 4, channels_last=True)
    coords = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float()
    coords = coords[None].repeat(N, 1, 1, 1).contiguous().to(cuda)
    out = db.corr_pyramid_lookup(pyr, coords, 3)
    centre = out[:, 24].float()                       # level 0, offsets (0,0): channel 3*7+3
    want = (f.float() / 4).pow(2).sum(-1)
    assert torch.allclose(centre, want, rtol=2e-3, atol=2e-2)
    assert (out[:, :49].float().argmax(1) == 24).float().mean() > 0.99


@pytest.mark.gpu
def test_volume_pool_matches_corrblock_through_edge_changes(cuda):
    """CorrVolumePool (resident slots + slot-indexed build/lookup) == CorrBlock (dense pyramid, cat / index) after
    adding, dropping and re-adding edges; bit-exact (same kernels, different addressing)."""
    from pvo_amd.modules.corr import CorrBlock, CorrVolumePool
    g = torch.Generator().manual_seed(3)
    H, W, C = 16, 24, 64
    f = torch.randn(6, H, W, C, generator=g).half().to(cuda)
    ii, jj = [0, 1, 2, 3, 4], [1, 2, 3, 4, 5]
    pool = CorrVolumePool(8, H, W, cuda)
    pool.add(f[ii], f[jj])
    blk = CorrBlock(f[ii][None], f[jj][None], channels_last=True)
    keep = [True, False, True, True, False]
    pool.keep(keep)
    blk = blk[torch.tensor(keep, device=cuda)]
    pool.add(f[[5, 0]], f[[0, 3]])
    blk = blk.cat(CorrBlock(f[[5, 0]][None], f[[0, 3]][None], channels_last=True))
    assert len(pool) == 5 and sorted(pool.slots + pool.free) == list(range(8))
    coords = (torch.rand(1, 5, H, W, 2, generator=g) * 24 - 2).to(cuda)
    a, b = pool(coords, channels_last=True), blk(coords, channels_last=True)
    assert a.shape == b.shape == (1, 5, 196, H, W) and torch.equal(a, b)


def _tile(level, th, tw):
    """row-major [..., h, w] -> 8x8-tiled [..., th, tw, 8, 8] (zero padded)"""
    h, w = level.shape[-2:]
    pad = torch.zeros(level.shape[:-2] + (th * 8, tw * 8), dtype=level.dtype, device=level.device)
    pad[..., :h, :w] = level
    return pad.unflatten(-2, (th, 8)).unflatten(-1, (tw, 8)).transpose(-3, -2).contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,W", [(48, 64), (8, 64), (24, 128), (30, 101), (16, 24), (20, 40), (11, 19), (47, 156)])
def test_tiled_pool_layout_is_bit_identical_to_row_major(cuda, dtype, H, W):
    """the 8x8-tiled build writes exactly the row-major pyramid's values at the tiled addresses, and the tiled
    lookup returns exactly what the row-major lookup returns (out-of-range, negative and border coordinates)"""
    from pvo_amd import droid_backends as db
    from pvo_amd.modules.corr import CorrVolumePool
    g = torch.Generator().manual_seed(H * 1000 + W)
    C, N = 64, 3
    f1 = torch.randn(N, H, W, C, generator=g).to(dtype).to(cuda)
    f2 = torch.randn(N, H, W, C, generator=g).to(dtype).to(cuda)
    ref = db.corr_build(f1, f2, 4, channels_last=True)
    pool = CorrVolumePool(5, H, W, cuda, dtype)
    assert pool.tiled
    for lv in pool.levels:
        lv.view(torch.int16).fill_(0x7e01)                      # NaN bit pattern in every padding element
    pool.add(f1, f2)
    slots = pool.slots
    for l in range(4):
        th, tw = db.tiled_level_shape(H, W, l)[:2]
        want = _tile(ref[l], th, tw)
        got = pool.levels[l][slots]
        hl, wl = H >> l, W >> l
        inside = _tile(torch.ones(hl, wl, device=cuda), th, tw).bool()
        assert torch.equal(got[..., inside].view(torch.int16), want[..., inside].view(torch.int16)), l
    coords = torch.rand(1, N, H, W, 2, generator=g) * torch.tensor([W + 12.0, H + 12.0]) - 6.0
    coords[0, 0, 0, :4] = torch.tensor([[-3.0, -3.0], [W + 2.5, H + 2.5], [0.0, 0.0], [W - 1.0, H - 1.0]])
    coords = coords.to(cuda)
    for cl in (False, True):
        a = pool(coords, channels_last=cl)[0]
        b = db.corr_pyramid_lookup(ref, coords[0].contiguous(), 3, channels_last=cl)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.gpu
def test_tiled_entry_points_reject_unsupported_shapes(cuda):
    from pvo_amd import droid_backends as db
    from pvo_amd.modules.corr import CorrVolumePool
    assert db.tiled_supported(30, 101, torch.float16) and not db.tiled_supported(48, 64, torch.float32)
    assert not db.tiled_supported(6, 24, torch.float16)       # level 3 would be empty
    assert not CorrVolumePool(2, 16, 24, cuda, torch.float32).tiled          # falls back to row-major planes
    f = torch.randn(1, 6, 24, 64, device=cuda).half()
    lv = [torch.empty((1, 6, 24) + db.tiled_level_shape(6, 24, l), dtype=torch.half, device=cuda) for l in range(4)]
    with pytest.raises(db.PvoHipError):
        db.corr_build_tiled(f, f, lv, torch.zeros(1, dtype=torch.int32, device=cuda))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,W", [(48, 64), (8, 64), (30, 101), (16, 24), (47, 156)])
def test_fused_lookup_encoder_matches_lookup_then_conv(cuda, dtype, H, W):
    """pvo_corr_lookup_encode_tiled == relu(conv1x1(lookup) + b): the lookup half is bit-exact (tested above), the
    encoder half is an fp32-accumulated GEMM rounded once to the 16-bit type"""
    import torch.nn.functional as F
    from pvo_amd import droid_backends as db
    from pvo_amd.modules.corr import CorrVolumePool
    g = torch.Generator().manual_seed(H + W)
    C, N = 64, 3
    f1 = torch.randn(N, H, W, C, generator=g).to(dtype).to(cuda)
    f2 = torch.randn(N, H, W, C, generator=g).to(dtype).to(cuda)
    pool = CorrVolumePool(4, H, W, cuda, dtype)
    pool.add(f1, f2)
    coords = (torch.rand(1, N, H, W, 2, generator=g) * torch.tensor([W + 8.0, H + 8.0]) - 4.0).to(cuda)
    w = (torch.randn(128, 196, 1, 1, generator=g) * 0.05).to(dtype).to(cuda)
    b = torch.randn(128, generator=g).to(cuda)
    got = pool.encoded(coords, db.corr_encoder_weights(w, dtype), b)
    corr = pool(coords, channels_last=True)[0]
    ref = torch.relu(F.conv2d(corr.float(), w.float(), b))
    assert got.shape == (N, 128, H, W) and got.is_contiguous(memory_format=torch.channels_last)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert torch.allclose(got.float(), ref, atol=tol, rtol=tol)
    assert (got.float() - ref).abs().mean().item() < tol / 8


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(48, 64), (30, 101), (47, 156)])
def test_tiled_and_fused_lookup_against_the_oracle_at_full_map_size(cuda, H, W):
    """The kernels the bench times - the tiled pool's build, its lookup, and the lookup fused with corr_encoder[0] -
    DIRECTLY against the CPU oracle at the S-B (48x64), S-A (30x101, VKITTI2 through test_vo.py) and S-1 (47x156 = 376x1248 / 8,
    VKITTI2 through test_vo2.py - BASELINE configs[0]) map sizes, C = 128: the
    oracle's lookup of the volume the pool holds
    (bit-exact; the volume itself is compared with oracle_corr_build to one unit in the last place, as for the row-major
    build above), and the encoder layer as an fp64 GEMM of the oracle's lookup."""
    from oracle import oracle as O
    from pvo_amd import droid_backends as db
    from pvo_amd.modules.corr import CorrVolumePool
    C, N = 128, 2
    g = torch.Generator().manual_seed(11)
    f1 = torch.randn(N, H, W, C, generator=g).half()
    f2 = torch.randn(N, H, W, C, generator=g).half()
    base = torch.stack(torch.meshgrid(torch.arange(W).float(), torch.arange(H).float(), indexing="xy"), -1)
    coords = base[None] + torch.randn(N, H, W, 2, generator=g) * 5.0
    coords[0, 0, :4] = torch.tensor([[-3.0, -3.0], [W + 2.5, H + 2.5], [0.0, 0.0], [W - 1.0, H - 1.0]])
    pool = CorrVolumePool(3, H, W, cuda, torch.float16)
    assert pool.tiled
    pool.add(f1.to(cuda), f2.to(cuda))
    # the pool's volumes, un-tiled on the host
    pyr = []
    for l in range(4):
        hl, wl = H >> l, W >> l
        t = pool.levels[l][pool.slots].cpu()                                              # [N,H,W,th,tw,8,8]
        pyr.append(t.transpose(-3, -2).reshape(N, H, W, t.shape[3] * 8, t.shape[4] * 8)[..., :hl, :wl].contiguous().numpy())
    ref_pyr = O.corr_build(f1[:1].permute(0, 3, 1, 2).contiguous().numpy(), f2[:1].permute(0, 3, 1, 2).contiguous().numpy(), 4)
    r0 = ref_pyr[0].astype(np.float64)      # one fp16 ulp + the fp32 accumulation-order noise of a 128-term sum
    assert np.all(np.abs(pyr[0][:1].astype(np.float64) - r0) <= 1.01 * 2.0 ** -10 * np.maximum(np.abs(r0), 2.0 ** -14) + 1e-4)
    for l in range(1, 4):                                                                 # pooled levels: exact functions of the stored level below
        assert np.array_equal(pyr[l].view(np.uint16), _pool_exact(pyr[l - 1], np.float16).view(np.uint16)), l
    want = O.corr_pyramid_lookup(pyr, coords.numpy(), 3)                                   # [N,196,H,W] fp16
    got = pool(coords[None].to(cuda))[0]
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))       # build + tiled lookup: bit-exact
    w = (torch.randn(128, 196, 1, 1, generator=g) * 0.05).half()
    b = torch.randn(128, generator=g)
    enc = pool.encoded(coords[None].to(cuda), db.corr_encoder_weights(w.to(cuda), torch.float16), b.to(cuda))
    ref = torch.relu(torch.einsum("oc,nchw->nohw", w[:, :, 0, 0].double(), torch.from_numpy(want.astype(np.float64))) + b.double().view(1, -1, 1, 1))
    err = (enc.cpu().double() - ref).abs()
    assert err.max().item() < 4e-3 * max(1.0, ref.abs().max().item()) and err.mean().item() < 5e-4
