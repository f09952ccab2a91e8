"""Fixtures produced by the reference's KERNEL TEXT (tests/golden/gen_kernel_golden.py: line ranges of
correlation_kernels.cu / droid_kernels.cu compiled on the host behind a shim, in the build container; data only is
committed).  CPU: the oracle against them.  GPU: the HIP kernels, through the C ABI, against them.

Integer / index / per-pixel work and the 16-bit lookup: bit for bit.  fp32 sums whose ORDER the reference fixes by its
block reduction (256 strided partials + a tree) and the build fixes differently: relative 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


@pytest.fixture(scope="module")
def corr():
    return np.load(os.path.join(G, "corr_lookup_kernel.npz"))


@pytest.fixture(scope="module")
def geom():
    return np.load(os.path.join(G, "geom_kernels.npz"))


@pytest.fixture(scope="module")
def alt():
    return np.load(os.path.join(G, "altcorr_kernel.npz"))


@pytest.fixture(scope="module")
def bak():
    return np.load(os.path.join(G, "ba_assemble_kernel.npz"))


CASES = "abcd"


# ---------------------------------------------------------------------------------------------------- lookup, CPU
@pytest.mark.parametrize("case", CASES)
def test_oracle_lookup_forward_is_the_kernel_text_bit_for_bit(corr, case):
    vol, co, r = corr[case + "_volume"], corr[case + "_coords"], int(corr[case + "_radius"])
    for tag, contract in (("fma", True), ("nofma", False)):
        got = O.corr_index_forward(vol, co, r, contract=contract)
        assert np.array_equal(_bits(got), _bits(corr[case + "_fwd_f32_" + tag])), tag
    got = O.corr_index_forward(vol.astype(np.float16), co, r)
    assert got.dtype == np.float16 and np.array_equal(_bits(got), _bits(corr[case + "_fwd_f16"]))


@pytest.mark.parametrize("case", CASES)
def test_oracle_lookup_backward_is_the_kernel_text_bit_for_bit(corr, case):
    vol, co, r, grad = corr[case + "_volume"], corr[case + "_coords"], int(corr[case + "_radius"]), corr[case + "_grad"]
    for tag, contract in (("fma", True), ("nofma", False)):
        got = O.corr_index_backward(vol.shape, co, grad, r, contract=contract)
        assert np.array_equal(_bits(got), _bits(corr[case + "_bwd_f32_" + tag])), tag
    got = O.corr_index_backward(vol.shape, co, grad.astype(np.float16), r)
    assert np.array_equal(_bits(got), _bits(corr[case + "_bwd_f16"]))


def test_lookup_fixture_exercises_what_it_claims(corr):
    # borders: some windows entirely outside (all-zero outputs), some partly; FMA contraction visible in fp32
    assert (corr["c_fwd_f32_fma"] == 0).mean() > 0.5 and (corr["a_fwd_f32_fma"] != 0).mean() > 0.5
    assert (corr["a_fwd_f32_fma"] != corr["a_fwd_f32_nofma"]).any()
    assert int(corr["c_radius"]) == 2 and int(corr["a_radius"]) == 3


# ---------------------------------------------------------------------------------------------------- geometry, CPU
@pytest.mark.parametrize("case", "ab")
def test_oracle_geometry_kernels_against_the_kernel_text(geom, case):
    poses, disps, intr, ii, jj = (geom[case + "_" + k] for k in ("poses", "disps", "intr", "ii", "jj"))
    co, va = O.projmap(poses, disps, intr, ii, jj)
    assert np.array_equal(_bits(co), _bits(geom[case + "_projmap_coords"]))
    assert np.array_equal(va, geom[case + "_projmap_valid"])
    assert np.array_equal(_bits(O.iproj(poses, disps, intr)), _bits(geom[case + "_iproj"]))
    for t in (0.005, 0.05):
        th = np.full(len(poses), t, np.float32)
        got = O.depth_filter(poses, disps, intr, np.arange(len(poses)), th)
        ref = geom[case + "_depth_filter_t%g" % t]
        assert np.array_equal(got, ref) and ref.max() >= 1      # integer counts: exact
    for beta in (0.3, 1.0, 0.0):
        got, ref = O.frame_distance(poses, disps, intr, ii, jj, beta), geom[case + "_frame_distance_beta%g" % beta]
        assert np.array_equal(got >= 1000, ref >= 1000)          # the valid / total < 0.75 rule (droid_kernels.cu:634)
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)   # the oracle sums in fp64, the kernel in an fp32 tree
    if case == "b":
        assert (geom["b_frame_distance_beta0.3"] >= 1000).sum() >= 2


def test_oracle_geometry_against_the_contracted_kernel_text(geom):
    """the same kernels with the multiply-adds contracted as nvcc does by default ("_fma" arrays; g++ chose the contractions):
    the oracle states every rounding, so it equals the uncontracted run bit for bit (above) and the contracted one to a few
    ulp; the integer outputs (validity masks, depth-filter counts) are the same in both runs of the text"""
    for case in "ab":
        poses, disps, intr, ii, jj = (geom[case + "_" + k] for k in ("poses", "disps", "intr", "ii", "jj"))
        co, va = O.projmap(poses, disps, intr, ii, jj)
        ref = geom[case + "_projmap_coords_fma"]
        ok = np.isfinite(ref) & (np.abs(ref) < 1e6)
        np.testing.assert_allclose(co[ok], ref[ok], rtol=3e-6, atol=3e-5)
        assert np.array_equal(va, geom[case + "_projmap_valid_fma"])
        np.testing.assert_allclose(O.iproj(poses, disps, intr), geom[case + "_iproj_fma"], rtol=3e-6, atol=1e-5)
        for t in (0.005, 0.05):
            assert np.array_equal(geom[case + "_depth_filter_t%g" % t], geom[case + "_depth_filter_t%g_fma" % t])
        for beta in (0.3, 1.0, 0.0):
            a, b = geom[case + "_frame_distance_beta%g" % beta], geom[case + "_frame_distance_beta%g_fma" % beta]
            assert np.array_equal(a >= 1000, b >= 1000)
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)
    assert (geom["b_projmap_coords"] != geom["b_projmap_coords_fma"]).any()       # the two forms do differ


def test_oracle_retraction_is_the_kernel_text_bit_for_bit(geom):
    """pose_retr_kernel (droid_kernels.cu:877-910) run from its text in both DEFINED readings of expSE3's out-of-bounds
    xi[45] (:154): the local `float xi[6]` padded with zeros ("xi45_zero") and upstream's xi[5] ("xi5").  Identity update,
    the theta <= 1e-4 and theta^2 < 1e-8 branches, a rotation near pi and one about the z axis are among the rows."""
    poses, dx = geom["retr_poses"], geom["retr_dx"]
    for tag, z in (("xi45_zero", True), ("xi5", False)):
        got = O.pose_retr(poses, dx, 0, len(poses), xi45_zero=z)
        assert np.array_equal(_bits(got), _bits(geom["retr_out_" + tag])), tag
    assert np.array_equal(geom["retr_out_xi5"][0], poses[0])                      # the identity update
    d = np.abs(geom["retr_out_xi45_zero"] - geom["retr_out_xi5"]).max(axis=1)
    assert d[4] > 1e-2 and d[0] == 0                                               # the read matters where xi[5] is large
    np.testing.assert_allclose(geom["retr_out_xi5_fma"], geom["retr_out_xi5"], rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------------------- BA, CPU
@pytest.mark.parametrize("case", "ab")
def test_oracle_ba_assembly_per_pixel_rows_are_the_kernel_text(bak, case):
    a = {k: bak[case + "_" + k] for k in ("poses", "disps", "intr", "targets", "weights", "ii", "jj")}
    r = O.ba_assemble(a["poses"], a["disps"], a["intr"], a["targets"], a["weights"], a["ii"], a["jj"])
    for k in ("Eii", "Eij", "Cii", "bz"):
        assert np.array_equal(_bits(r[k]), _bits(bak[case + "_" + k])), k     # fp32 per pixel: bit for bit
    for k in ("Hs", "vs"):
        ref = bak[case + "_" + k]
        assert np.abs(r[k] - ref).max() <= 1e-6 * np.abs(ref).max(), k         # 90 sums: fp64 here, fp32 tree there
    if case == "b":
        assert (bak["b_Cii"] == 0).any()                                     # pixels behind MIN_DEPTH took part


@pytest.mark.parametrize("case", "ab")
def test_oracle_ba_step_against_the_kernel_text_chain(bak, case):
    """one Gauss-Newton step: reduced system, dx, dz (every kernel of ba_cuda is the reference's text; the host code
    between them restated in the generator; solve = dense fp64 Cholesky)"""
    a = {k: bak[case + "_" + k] for k in ("poses", "disps", "intr", "targets", "weights", "ii", "jj", "eta")}
    P = len(a["poses"])
    r = O.ba(a["poses"], a["disps"], a["intr"], a["targets"], a["weights"], a["eta"], a["ii"], a["jj"], 1, P, 1, 1e-4, 0.1,
             want_sys=True)
    n6 = 6 * (P - 1)
    A, b = r["sys"][:n6 * n6].reshape(n6, n6), r["sys"][n6 * n6:]
    assert np.abs(A - bak[case + "_step_sysA"]).max() <= 2e-6 * np.abs(A).max()
    assert np.abs(b - bak[case + "_step_sysb"]).max() <= 2e-6 * np.abs(b).max()
    dx, dz = bak[case + "_step_dx"], bak[case + "_step_dz"]
    # case b is ill-conditioned on purpose (a blob at disparity 14: cond(A - S) = 5e4), the fp32 sums' 5e-7 shows in dx
    tol_dx, tol_dz = (1e-6, 1e-6) if case == "a" else (1e-3, 2e-3)
    assert np.abs(r["dx"] - dx).max() <= tol_dx * np.abs(dx).max()
    assert np.abs(r["dz"] - dz).max() <= tol_dz * np.abs(dz).max()
    assert r["K"] == len(bak[case + "_step_kx"]) and not r["failed"]
    # the end of the iteration from the text as well: disp_retr_kernel (:912-925) and pose_retr_kernel (:877-910, xi[5] reading)
    kx = bak[case + "_step_kx"]
    want_d, want_p = bak[case + "_step_disps"], bak[case + "_step_poses_xi5"]
    assert np.abs(r["disps"] - want_d).max() <= tol_dz * max(np.abs(dz).max(), 1e-3)
    assert np.array_equal(np.delete(r["disps"], kx, axis=0), np.delete(a["disps"], kx, axis=0))       # other frames untouched
    assert np.abs(r["poses"] - want_p).max() <= 2 * tol_dx * max(np.abs(dx).max(), 1e-3) and np.array_equal(r["poses"][0], a["poses"][0])
    # the retraction alone, on the text's own dx: bit for bit in both readings
    for tag, z in (("xi45_zero", True), ("xi5", False)):
        assert np.array_equal(_bits(O.pose_retr(a["poses"], dx, 1, P, xi45_zero=z)), _bits(bak[case + "_step_poses_" + tag])), tag
    # contraction (nvcc's default) moves the step by fp32 rounding only
    assert np.abs(bak[case + "_step_dx_fma"] - dx).max() <= (1e-5 if case == "a" else 2e-3) * np.abs(dx).max()
    for k in ("Eii", "Eij", "Cii", "bz"):
        ref, fma = bak[case + "_" + k], bak[case + "_" + k + "_fma"]
        assert np.abs(fma - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0), k


# ==================================================================================================== GPU (C ABI)


# ---------------------------------------------------------------------------------------------------- altcorr, CPU
@pytest.mark.parametrize("case", "abc")
def test_oracle_altcorr_forward_is_the_kernel_text_bit_for_bit(alt, case):
    """altcorr_kernel.cu:27-149 with the contraction nvcc applies by default (dot product AND the bilinear scatter are
    fused multiply-adds); the uncontracted run of the same text differs by rounding only"""
    f1, f2, co, r = alt[case + "_fmap1"], alt[case + "_fmap2"], alt[case + "_coords"], int(alt[case + "_radius"])
    got = O.altcorr_forward(f1, f2, co, r)
    assert np.array_equal(_bits(got), _bits(alt[case + "_fwd_fma"]))
    assert np.abs(got - alt[case + "_fwd_nofma"]).max() <= 4e-6 * np.abs(got).max()
    assert (alt[case + "_fwd_fma"] != alt[case + "_fwd_nofma"]).any()


@pytest.mark.parametrize("case", "bc")
def test_oracle_altcorr_backward_against_the_kernel_text(alt, case):
    """:152-286; fmap2's gradient is summed by atomicAdd in an order the text does not fix (the oracle sums in fp64)"""
    f1, f2, co, r = alt[case + "_fmap1"], alt[case + "_fmap2"], alt[case + "_coords"], int(alt[case + "_radius"])
    g1, g2 = O.altcorr_backward(f1, f2, co, alt[case + "_grad"], r)
    for got, want in ((g1, alt[case + "_bwd_fmap1_fma"]), (g2, alt[case + "_bwd_fmap2_fma"])):
        assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_lookup_forward_and_backward_are_the_kernel_text_bit_for_bit(cuda, corr, case):
    from pvo_amd import droid_backends as db
    vol, co, r = corr[case + "_volume"], corr[case + "_coords"], int(corr[case + "_radius"])
    grad = corr[case + "_grad"]
    v, c, g = (torch.from_numpy(x).to(cuda) for x in (vol, co, grad))
    out = db.corr_index_forward(v, c, r)[0]
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(corr[case + "_fwd_f32_fma"]))       # nvcc contracts: fma
    out = db.corr_index_forward(v.half(), c, r)[0]
    assert out.dtype == torch.float16 and np.array_equal(_bits(out.cpu().numpy()), _bits(corr[case + "_fwd_f16"]))
    out = db.corr_index_backward(v, c, g, r)[0]
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(corr[case + "_bwd_f32_fma"]))
    out = db.corr_index_backward(v.half(), c, g.half(), r)[0]
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(corr[case + "_bwd_f16"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", "ab")
def test_hip_fused_pyramid_lookup_equals_the_kernel_text_per_level(cuda, corr, case):
    """the one-launch pyramid lookup (what the product runs) on a pyramid whose level 0 is the fixture volume: its
    first 49 channels are the reference kernel's output for that level"""
    from pvo_amd import droid_backends as db
    if int(corr[case + "_radius"]) != 3:
        pytest.skip("fused form is radius 3")
    vol, co = corr[case + "_volume"], corr[case + "_coords"]
    N, h1, w1, h2, w2 = vol.shape
    g = np.random.default_rng(5)
    pyr = [vol.astype(np.float16)] + [g.standard_normal((N, h1, w1, h2 >> l, w2 >> l)).astype(np.float16)
                                      for l in (1, 2, 3)]
    c_nhw2 = np.ascontiguousarray(np.transpose(co, (0, 2, 3, 1)))
    got = db.corr_pyramid_lookup([torch.from_numpy(p).to(cuda) for p in pyr], torch.from_numpy(c_nhw2).to(cuda), 3)
    got = got.cpu().numpy().reshape(N, 4, 7, 7, h1, w1)[:, 0]
    assert np.array_equal(_bits(got), _bits(corr[case + "_fwd_f16"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", "ab")
def test_hip_geometry_kernels_against_the_kernel_text(cuda, geom, case):
    from pvo_amd import droid_backends as db
    poses, disps, intr, ii, jj = (torch.from_numpy(geom[case + "_" + k]).to(cuda) for k in ("poses", "disps", "intr", "ii", "jj"))
    # geom.hip is compiled without multiply-add contraction (pvo_amd/build.py EXTRA_FLAGS): the kernels perform the roundings the
    # reference's text states, so everything per pixel is bit for bit the uncontracted run of that text - the integer outputs
    # (validity, the depth filter's counts) included, as this tier asks of integer work
    co, va = db.projmap(poses, disps, intr, ii, jj)
    assert np.array_equal(_bits(co.cpu().numpy()), _bits(geom[case + "_projmap_coords"]))
    assert np.array_equal(va.cpu().numpy(), geom[case + "_projmap_valid"])
    assert np.array_equal(_bits(db.iproj(poses, disps, intr).cpu().numpy()), _bits(geom[case + "_iproj"]))
    P = poses.shape[0]
    for t in (0.005, 0.05):
        th = torch.full((P,), t, device=cuda)
        got = db.depth_filter(poses, disps, intr, torch.arange(P, device=cuda), th).cpu().numpy()
        assert np.array_equal(got, geom[case + "_depth_filter_t%g" % t])
    for beta in (0.3, 1.0, 0.0):
        got = db.frame_distance(poses, disps, intr, ii, jj, beta).cpu().numpy()
        ref = geom[case + "_frame_distance_beta%g" % beta]
        assert np.array_equal(got >= 1000, ref >= 1000)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", "ab")
def test_hip_ba_step_against_the_kernel_text_chain(cuda, bak, case):
    from pvo_amd import droid_backends as db
    a = {k: torch.from_numpy(bak[case + "_" + k]).to(cuda) for k in ("poses", "disps", "intr", "targets", "weights", "ii", "jj", "eta")}
    P = a["poses"].shape[0]
    status = torch.zeros(4, dtype=torch.int32, device=cuda)
    dx, dz = db.ba(a["poses"], a["disps"], a["intr"], a["targets"], a["weights"], a["eta"], a["ii"], a["jj"], 1, P, 1, 1e-4, 0.1,
                   False, status=status)
    rdx, rdz = bak[case + "_step_dx"], bak[case + "_step_dz"]
    tol_dx, tol_dz = (1e-4, 1e-4) if case == "a" else (2e-3, 4e-3)     # b: cond 5e4, see the CPU test
    assert np.abs(dx.cpu().numpy() - rdx).max() <= tol_dx * max(np.abs(rdx).max(), 1e-3)
    assert np.abs(dz.cpu().numpy() - rdz).max() <= tol_dz * max(np.abs(rdz).max(), 1e-3)
    assert status.cpu().numpy()[0] == 0
    # ... and the state the iteration leaves behind against pose_retr_kernel / disp_retr_kernel run from their text on the
    # text's own dx / dz (the library retracts with xi[5], upstream's reading of droid_kernels.cu:154)
    want_p, want_d = bak[case + "_step_poses_xi5"], bak[case + "_step_disps"]
    assert np.abs(a["poses"].cpu().numpy() - want_p).max() <= 2 * tol_dx * max(np.abs(rdx).max(), 1e-3)
    assert np.abs(a["disps"].cpu().numpy() - want_d).max() <= tol_dz * max(np.abs(rdz).max(), 1e-3)
    assert torch.equal(a["poses"][0].cpu(), torch.from_numpy(bak[case + "_poses"][0]))


@pytest.mark.gpu
def test_hip_retraction_against_the_kernel_text(cuda, geom):
    """the SE3 retraction the product uses elsewhere (pvo_amd.geom.se3, HIP kernels of se3_ops.hip: lietorch's formulas) against
    pose_retr_kernel run from its text with xi[5]: the two exponentials are different closed forms of the same map"""
    from pvo_amd.geom.se3 import SE3
    poses, dx = torch.from_numpy(geom["retr_poses"]).to(cuda), torch.from_numpy(geom["retr_dx"]).to(cuda)
    got = SE3(poses).retr(dx).data.cpu().numpy()
    want = geom["retr_out_xi5"]
    sign = np.sign((got[:, 3:] * want[:, 3:]).sum(1, keepdims=True))               # q and -q are the same rotation
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=0, atol=5e-6)
    np.testing.assert_allclose(got[:, 3:] * sign, want[:, 3:], rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", "abc")
def test_hip_altcorr_against_the_kernel_text(cuda, alt, case):
    """the HIP kernels sum the 128 channels in another order (lanes of a wave, not 32-channel slabs): fp32 rounding"""
    from pvo_amd import droid_backends as db
    f1, f2, co, r = alt[case + "_fmap1"], alt[case + "_fmap2"], alt[case + "_coords"], int(alt[case + "_radius"])
    t = lambda a: torch.from_numpy(a).to(cuda)
    got, = db.altcorr_forward(t(f1), t(f2), t(co), r)
    want = alt[case + "_fwd_fma"]
    assert np.abs(got.cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()
    if case + "_grad" in alt:
        g1, g2, gc = db.altcorr_backward(t(f1), t(f2), t(co), t(alt[case + "_grad"]), r)
        for g, w in ((g1, alt[case + "_bwd_fmap1_fma"]), (g2, alt[case + "_bwd_fmap2_fma"])):
            assert np.abs(g.cpu().numpy() - w).max() <= 2e-6 * np.abs(w).max()
        assert not gc.any()
