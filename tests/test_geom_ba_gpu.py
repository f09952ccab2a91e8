"""GPU parity: reprojection kernels and the device-resident BA against the CPU oracle
(restated droid_kernels.cu, pinned by the reference's Python BA in tests/test_ba_oracle.py).
Floating point: tolerance 1e-4 on poses / disparities / flow (BASELINE.json north_star),
1e-5 relative on per-pixel maps."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

# subprocess prelude: the library's test hook (include/pvo_hip.h pvo_debug_config) from "knob=value" arguments
_KNOBS = ("import sys; sys.path.insert(0, %r); from pvo_amd import droid_backends as _dbk; "
          "[_dbk.debug_config(kv.split('=')[0], kv.split('=')[1]) for kv in sys.argv[2:]]; ") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")


def _scene(seed, P, ht, wd, radius=3, t0=1, noise=0.1, nframes=None):
    """Synthetic window in the S-B recipe, any size."""
    from pvo_amd.geom.se3 import SE3
    g = torch.Generator().manual_seed(seed)
    F = nframes or P
    intr = torch.tensor([wd * 0.625, wd * 0.625, wd / 2.0, ht / 2.0])
    xi = torch.tensor([0.05, 0.0, 0.02, 0.0, 0.01, 0.0])
    poses_gt = torch.stack([SE3.exp(k * xi).data for k in range(F)], 0)
    low = torch.rand(1, 1, 6, 8, generator=g) * 0.8 + 0.2
    disps_gt = torch.nn.functional.interpolate(low, size=(ht, wd), mode="bilinear", align_corners=True)[0, 0]
    disps_gt = disps_gt[None].repeat(F, 1, 1)
    ii, jj = [], []
    for i in range(P):
        for j in range(P):
            if i != j and abs(i - j) <= radius:
                ii.append(i); jj.append(j)
    ii, jj = torch.tensor(ii), torch.tensor(jj)
    intr_all = intr[None].repeat(F, 1)
    c, _ = O.reproject(poses_gt.numpy(), disps_gt.numpy(), intr_all.numpy(), ii.numpy(), jj.numpy())
    E = ii.shape[0]
    target = torch.from_numpy(c) + noise * torch.randn(E, ht, wd, 2, generator=g)
    weight = torch.rand(E, ht, wd, 2, generator=g)
    poses0 = torch.stack([poses_gt[max(k - 1, 0)] for k in range(F)], 0)
    disps0 = torch.ones(F, ht, wd)
    eta = torch.full((P, ht, wd), 1e-4) + 0.01 * torch.rand(P, ht, wd, generator=g)
    return dict(intr=intr, poses=poses0, disps=disps0, target=target.permute(0, 3, 1, 2).contiguous(),
                weight=weight.permute(0, 3, 1, 2).contiguous(), eta=eta, ii=ii, jj=jj, t0=t0, t1=P,
                poses_gt=poses_gt, disps_gt=disps_gt)


def _run_ba(s, cuda, iters, lm=1e-4, ep=0.1, motion_only=False, eta=None):
    from pvo_amd import droid_backends as db
    poses, disps = s["poses"].clone().to(cuda), s["disps"].clone().to(cuda)
    eta_t = (s["eta"] if eta is None else eta)
    status = torch.zeros(4, dtype=torch.int32, device=cuda)
    dx, dz = db.ba(poses, disps, s["intr"].to(cuda), s["target"].to(cuda), s["weight"].to(cuda),
                   None if motion_only else eta_t.to(cuda), s["ii"].to(cuda), s["jj"].to(cuda),
                   s["t0"], s["t1"], iters, lm, ep, motion_only, status=status)
    return poses.cpu().numpy(), disps.cpu().numpy(), dx.cpu().numpy(), dz.cpu().numpy(), status.cpu().numpy()


@pytest.mark.parametrize("cfg", [(5, 8, 10, 2, 1), (4, 6, 9, 2, 2), (8, 12, 16, 3, 1), (6, 11, 13, 5, 1)])
@pytest.mark.parametrize("iters", [1, 2])
def test_ba_matches_oracle(cuda, cfg, iters):
    P, ht, wd, radius, t0 = cfg
    s = _scene(P * 100 + ht, P, ht, wd, radius, t0)
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), t0, P, iters, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, iters)
    assert status[0] == 0 and status[1] == want["K"] and status[2] == 0
    assert np.abs(poses - want["poses"]).max() < 1e-4
    assert np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4
    assert np.abs(dz - want["dz"]).max() < 1e-4


@pytest.mark.parametrize("P,closure", [(30, False), (30, True), (23, True), (22, True), (64, "many")])
def test_ba_long_window_envelope_cholesky_matches_oracle(cuda, P, closure):
    """A long keyframe chain (block-banded reduced pose system; with `closure` loop-closure edges that create fill inside
    the envelope): the envelope-form Cholesky of ba_solve_kernel against the oracle's dense fp64 solve - dense in LDS at
    P = 22 (21 free poses), compact envelope blocks in LDS at 23 and 30 (ba_env_kernel + ba_prepare_kernel), and at P = 64
    with ten frames closing back to frame 1 an envelope that does not fit LDS: the dense global-memory path"""
    ht, wd = 8, 10
    s = _scene(P * 7 + int(bool(closure)), P, ht, wd, 2, 1)
    if closure:
        from pvo_amd.geom.se3 import SE3
        if closure == "many":
            far = list(range(P - 12, P - 2))
            extra_i, extra_j = torch.tensor([1] * len(far) + far), torch.tensor(far + [1] * len(far))
        else:
            extra_i, extra_j = torch.tensor([1, P - 2, 3, P - 5]), torch.tensor([P - 2, 1, P - 5, 3])
        ii, jj = torch.cat([s["ii"], extra_i]), torch.cat([s["jj"], extra_j])
        F = P
        c, _ = O.reproject(s["poses_gt"].numpy(), s["disps_gt"].numpy(), s["intr"][None].repeat(F, 1).numpy(), extra_i.numpy(), extra_j.numpy())
        g = torch.Generator().manual_seed(5)
        ne = extra_i.shape[0]
        t_extra = (torch.from_numpy(c) + 0.1 * torch.randn(ne, ht, wd, 2, generator=g)).permute(0, 3, 1, 2)
        s = dict(s, ii=ii, jj=jj, target=torch.cat([s["target"], t_extra]).contiguous(),
                 weight=torch.cat([s["weight"], 0.2 * torch.rand(ne, 2, ht, wd, generator=g)]).contiguous())
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert status[0] == 0 and status[1] == want["K"]
    assert np.abs(poses - want["poses"]).max() < 1e-4
    assert np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4


@pytest.mark.parametrize("radius,ht,wd", [(7, 8, 10), (13, 9, 11), (5, 9, 11)])
def test_ba_frames_with_many_neighbours_match_oracle(cuda, radius, ht, wd):
    """depth frames with more than nine free neighbours - a frontend window WITH its inactive edges (factor_graph.py:281-289,
    use_inactive) - have more than four 16-row tiles in the Schur kernel: the row-pass path (radius 7: six tiles, one pass per
    row tile; radius 13: ten tiles, passes of eight + two), on a map whose pixel count is and is not a multiple of four"""
    P = 26
    s = _scene(900 + radius, P, ht, wd, radius, 1)
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert status[0] == 0 and status[1] == want["K"] and status[2] == 0
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4 and np.abs(dz - want["dz"]).max() < 1e-4
    again = _run_ba(s, cuda, 2)
    assert np.array_equal(poses, again[0]) and np.array_equal(disps, again[1])          # fixed summation order: bitwise repeatable


@pytest.mark.parametrize("copies,ht,wd", [(3, 8, 10), (4, 9, 11)])
def test_ba_edges_that_repeat_a_frame_pair_match_oracle(cuda, copies, ht, wd):
    """A frontend window keeps its aged-out edges as inactive ones (factor_graph.py:281-289): the same (i, j) pair several times,
    each copy with its own target and weight.  The Schur kernel sums the row blocks of a frame's edges per TARGET pose before the
    products (`Mrg`): against the oracle, which treats every edge on its own; interleaved copies, a frame whose repeated target is
    the fixed pose 0, and one pair that appears once"""
    P = 10
    s = _scene(700 + copies, P, ht, wd, 2, 1)
    g = torch.Generator().manual_seed(copies)
    E = s["ii"].shape[0]
    order = torch.cat([torch.randperm(E, generator=g) for _ in range(copies)])[:-1]       # (the last copy of one edge is left out)
    tgt = s["target"][order] + 0.2 * torch.randn(order.shape[0], 2, ht, wd, generator=g)
    wgt = torch.rand(order.shape[0], 2, ht, wd, generator=g)
    s = dict(s, ii=s["ii"][order].contiguous(), jj=s["jj"][order].contiguous(), target=tgt.contiguous(), weight=wgt.contiguous())
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert status[0] == 0 and status[1] == want["K"] and status[2] == 0
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4 and np.abs(dz - want["dz"]).max() < 1e-4
    again = _run_ba(s, cuda, 2)
    assert np.array_equal(poses, again[0]) and np.array_equal(disps, again[1])


def _frontend_window(P=26, ht=30, wd=101, copies=4, seed=4242):
    """the BA of a real frontend update (factor_graph.py:281-291: the window's inactive edges enter the BA with the active ones;
    bench.py `sequence.ba_windows_sampled`: 22-26 poses, 48 active + 270-430 inactive edges, 15-19 neighbours per frame, the same
    frame pair several times, 30 x 101 maps = 3030 pixels = 5 chunks of 512 + 470): every |i - j| <= 2 pair `copies` times in
    interleaved order, each copy with its own target and weight, + the 48 active edges |i - j| <= 3 of the last ten frames"""
    s = _scene(seed, P, ht, wd, 2, 1)
    g = torch.Generator().manual_seed(seed + 1)
    E = s["ii"].shape[0]
    order = torch.cat([torch.randperm(E, generator=g) for _ in range(copies)])
    act = [(i, j) for i in range(P - 10, P) for j in range(P - 10, P) if i != j and abs(i - j) <= 3]
    assert len(act) == 48
    ai, aj = torch.tensor([a for a, _ in act]), torch.tensor([b for _, b in act])
    c, _ = O.reproject(s["poses_gt"].numpy(), s["disps_gt"].numpy(), s["intr"][None].repeat(P, 1).numpy(), ai.numpy(), aj.numpy())
    t_act = (torch.from_numpy(c) + 0.1 * torch.randn(48, ht, wd, 2, generator=g)).permute(0, 3, 1, 2)
    tgt = torch.cat([s["target"][order] + 0.2 * torch.randn(order.shape[0], 2, ht, wd, generator=g), t_act])
    wgt = torch.cat([0.3 * torch.rand(order.shape[0], 2, ht, wd, generator=g), torch.rand(48, 2, ht, wd, generator=g)])
    return dict(s, ii=torch.cat([s["ii"][order], ai]).contiguous(), jj=torch.cat([s["jj"][order], aj]).contiguous(),
                target=tgt.contiguous(), weight=wgt.contiguous())


def _two_virtual_ranks(s, cuda, iters=2, lm=1e-4, ep=0.1):
    """the same BA as two edge shards (by source keyframe) that take turns on the device, their integer systems added as the
    all-reduce would: (poses of rank 0, poses of rank 1, merged depth maps)"""
    from pvo_amd import droid_backends as db
    from pvo_amd.parallel import local_eta_rows, partition_by_source
    d = lambda t: t.to(cuda)
    owner, _ = partition_by_source(s["ii"].tolist(), 2)
    F, ht, wd = s["disps"].shape
    P = s["t1"] - s["t0"]
    shards = []
    for r in range(2):
        m = torch.tensor([o == r for o in owner])
        rows = local_eta_rows(s["ii"].tolist(), s["ii"][m].tolist(), s["t0"], s["t1"])
        sh = dict(ii=d(s["ii"][m].contiguous()), jj=d(s["jj"][m].contiguous()), target=d(s["target"][m].contiguous()),
                  weight=d(s["weight"][m].contiguous()), eta=d(s["eta"][rows].contiguous()),
                  poses=d(s["poses"].clone()), disps=d(s["disps"].clone()))
        sh["ws"] = db.ba_workspace(sh["ii"].shape[0], P, F, ht * wd, cuda)
        sh["sys"] = torch.zeros((6 * P) ** 2 + 6 * P, dtype=torch.int64, device=cuda)
        db.ba_plan(sh["ii"], sh["jj"], F, ht * wd, sh["eta"].shape[0], s["t0"], s["t1"], sh["ws"])
        shards.append(sh)
    for _ in range(iters):
        for sh in shards:
            db.ba_local(sh["poses"], sh["disps"], d(s["intr"]), sh["target"], sh["weight"], sh["eta"], sh["ii"], sh["jj"],
                        s["t0"], s["t1"], False, sh["sys"], sh["ws"])
        total = shards[0]["sys"] + shards[1]["sys"]
        for sh in shards:
            db.ba_finish(sh["poses"], sh["disps"], total.clone(), sh["ii"], sh["jj"], s["t0"], s["t1"], lm, ep, False, sh["ws"])
            sh["sys"].zero_()
    merged = d(s["disps"].clone()) + sum(sh["disps"] - d(s["disps"]) for sh in shards)
    return shards[0]["poses"], shards[1]["poses"], merged


@pytest.mark.parametrize("kind", ["radius8", "repeats", "radius13"])
def test_ba_of_a_real_frontend_window_matches_oracle_at_full_size(cuda, kind):
    """The BA the full sequence really runs: 26 poses at 30 x 101 - 512-pixel chunks (3030 = 5 x 512 + 470, not a
    multiple of four after chunking), four z-slices, the row-pass path, merged repeated targets, the packed row table, the
    partitioned / window pose solve - against the oracle at 1e-4, twice for bitwise repeatability, and once through two virtual
    ranks.  "radius8": 344 edges, 16 distinct neighbours per frame; "repeats": 440 edges, every inactive pair four times in
    interleaved order + 48 active edges (the shape `sequence.ba_windows_sampled` reports); "radius13": 494 edges, the frames in the
    middle of the window with 25 neighbours - 157 rows, more than the dense-window Schur form can stage in LDS: those frames stream
    their rows and issue their atomics directly while the frames at the ends (13 neighbours) take the staged, two-stage form, in
    one launch."""
    P, ht, wd = 26, 30, 101
    s = _frontend_window(P, ht, wd) if kind == "repeats" else _scene(8801, P, ht, wd, 8 if kind == "radius8" else 13, 1)
    assert s["ii"].shape[0] == {"radius8": 344, "repeats": 440, "radius13": 494}[kind]
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert status[0] == 0 and status[1] == want["K"] == P and status[2] == 0 and status[3] == 0
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4 and np.abs(dz - want["dz"]).max() < 1e-4
    again = _run_ba(s, cuda, 2)
    assert np.array_equal(poses, again[0]) and np.array_equal(disps, again[1])          # fixed summation order: bitwise repeatable
    p0, p1, merged = _two_virtual_ranks(s, cuda)
    assert torch.equal(p0, p1)                                                          # replicas bit-identical
    assert np.array_equal(p0.cpu().numpy(), poses)                                      # ... and equal to the whole graph on one GPU
    assert np.abs(merged.cpu().numpy() - disps).max() < 2e-6


def test_ba_with_a_wrong_eta_row_count_updates_nothing(cuda):
    """eta must have one row per depth frame (or one, broadcast).  K is found on the device, so the host cannot refuse the call:
    the mismatch is reported in the status words and the call changes neither poses nor depths (include/pvo_hip.h pvo_ba)"""
    P, ht, wd = 8, 12, 16
    s = _scene(31, P, ht, wd, 3, 1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2, eta=s["eta"][:5])
    assert status[0] == 1 and status[2] == 1 and status[1] == P
    assert np.array_equal(poses, s["poses"].numpy()) and np.array_equal(disps, s["disps"].numpy())
    assert not dx.any()


def test_ba_matches_oracle_at_SA_size(cuda):
    """S-A: 30x101 maps, 10 keyframes, 48 edges - the window shape of the reference's own driver"""
    P, ht, wd = 10, 30, 101
    s = _scene(77, P, ht, wd, 3, 1)
    assert s["ii"].shape[0] == 48
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert status[0] == 0 and status[1] == want["K"]
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4 and np.abs(dz - want["dz"]).max() < 1e-4


def test_ba_matches_oracle_at_S20_size(cuda):
    """S-20 (BASELINE.json configs[3]) at its full size: 64 keyframes, the 372 edges |i - j| <= 3, 48x64 maps, 63 free
    poses (the envelope solve in LDS) - the device BA against the oracle's dense fp64 solve"""
    P, ht, wd = 64, 48, 64
    s = _scene(2020, P, ht, wd, 3, 1)
    assert s["ii"].shape[0] == 372
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert status[0] == 0 and status[1] == want["K"] == 64
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4 and np.abs(dz - want["dz"]).max() < 1e-4


def test_ba_matches_reference_python_fixture_poses(cuda):
    """Poses after a native BA step equal the reference geom/ba.py result (the pose update is
    unaffected by EvT6x1's pose-0 skip); fixtures generated from /root/reference."""
    from pvo_amd import droid_backends as db
    d = dict(np.load(os.path.join(G, "ba_python_a.npz")))
    t0, P = int(d["fixedp"]), d["poses"].shape[0]
    tw = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2))).to(cuda)
    poses, disps = torch.from_numpy(d["poses"]).to(cuda), torch.from_numpy(d["disps"]).to(cuda)
    db.ba(poses, disps, torch.from_numpy(d["intr"]).to(cuda), tw(d["target"]), tw(d["weight"]),
          torch.from_numpy(d["eta"]).to(cuda), torch.from_numpy(d["ii"]).to(cuda), torch.from_numpy(d["jj"]).to(cuda),
          t0, P, 1, 1e-4, 0.1, False)
    assert np.abs(poses.cpu().numpy() - d["ba_poses_1"]).max() < 1e-4


def test_ba_motion_only_and_depth_only(cuda):
    s = _scene(7, 6, 9, 12, 2, 1)
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                None, s["ii"].numpy(), s["jj"].numpy(), 1, 6, 2, 1e-4, 0.1, motion_only=True)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2, motion_only=True)
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.array_equal(disps, s["disps"].numpy())
    assert dz.shape[0] == 0
    # P == 0: the 2-frame fixedp=2 case of test_vo2.py degenerates to a depth-only update
    s2 = _scene(8, 2, 7, 9, 1, 2)
    want = O.ba(s2["poses"].numpy(), s2["disps"].numpy(), s2["intr"].numpy(), s2["target"].numpy(),
                s2["weight"].numpy(), s2["eta"].numpy(), s2["ii"].numpy(), s2["jj"].numpy(), 2, 2, 1, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s2, cuda, 1)
    assert np.array_equal(poses, s2["poses"].numpy())
    assert np.abs(disps - want["disps"]).max() < 1e-5 and status[1] == want["K"] == 2


def test_ba_fixed_source_frame_and_broadcast_eta(cuda):
    """t0 = 2 with edges out of frames 0,1: their depths are optimised, their poses are not;
    a single eta row is broadcast (eta [1,ht,wd])."""
    s = _scene(9, 6, 8, 11, 2, 2)
    eta1 = s["eta"][:1].contiguous()
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                eta1.numpy(), s["ii"].numpy(), s["jj"].numpy(), 2, 6, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2, eta=eta1)
    assert want["K"] == 6 and dz.shape[0] == 6
    assert np.array_equal(poses[:2], s["poses"].numpy()[:2])
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4


@pytest.mark.parametrize("P,radius", [(64, 3), (40, 2), (31, 3)])
def test_partitioned_pose_solve_partitions_a_chain_and_matches_the_oracle(cuda, P, radius):
    """the pose solve beyond the dense LDS path (ba_solve_twin_kernel): a keyframe chain is eliminated from both ends by two
    workgroups, the separator last - the partition is reported, balanced, and the result is the oracle's (dense fp64 solve)"""
    from pvo_amd import droid_backends as db
    ht, wd = 8, 10
    s = _scene(P + radius, P, ht, wd, radius, 1)
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 1, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 1)
    m, sp = db.ba_last_partition(s["ii"].shape[0], P - 1, P, ht * wd, cuda)
    free = P - 1
    assert 2 <= m < sp <= free - 2 and sp - m == 2 * radius, (m, sp)            # separator: the 2 * radius poses coupled across the cut
    assert abs(m - (free - sp)) <= 1                                             # the two chains are as long as each other
    assert status[0] == 0 and status[1] == want["K"]
    assert np.abs(dx - want["dx"]).max() < 2e-6 * max(1.0, np.abs(want["dx"]).max() / 1e-2)
    assert np.abs(poses - want["poses"]).max() < 1e-5 and np.abs(disps - want["disps"]).max() < 1e-4


def test_partitioned_pose_solve_declines_a_wide_separator(cuda):
    """loop closures from the far end back to the first poses couple everything below the cut with the end of the chain: the
    separator would be wider than the exchange buffer allows, the solve stays one chain (and is still the oracle's)"""
    from pvo_amd import droid_backends as db
    P, ht, wd = 32, 8, 10            # (31 free poses: beyond the dense matrix-core solve, which takes windows up to 29)
    s = _scene(77, P, ht, wd, 2, 1)
    far = list(range(P - 14, P - 1))
    extra_i, extra_j = torch.tensor([1] * len(far) + far), torch.tensor(far + [1] * len(far))
    c, _ = O.reproject(s["poses_gt"].numpy(), s["disps_gt"].numpy(), s["intr"][None].repeat(P, 1).numpy(), extra_i.numpy(), extra_j.numpy())
    g = torch.Generator().manual_seed(5)
    ne = extra_i.shape[0]
    t_extra = (torch.from_numpy(c) + 0.1 * torch.randn(ne, ht, wd, 2, generator=g)).permute(0, 3, 1, 2)
    s = dict(s, ii=torch.cat([s["ii"], extra_i]), jj=torch.cat([s["jj"], extra_j]), target=torch.cat([s["target"], t_extra]).contiguous(),
             weight=torch.cat([s["weight"], 0.2 * torch.rand(ne, 2, ht, wd, generator=g)]).contiguous())
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 1, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 1)
    assert db.ba_last_partition(s["ii"].shape[0], P - 1, P, ht * wd, cuda) == (0, 0)
    assert status[0] == 0 and np.abs(poses - want["poses"]).max() < 1e-5


def test_partitioned_pose_solve_reports_a_non_spd_system(cuda):
    """a pivot fails in one of the two parts: both write a zero update, the status says so (no hang: the hand-over carries
    the failure)"""
    P = 40
    s = _scene(11, P, 8, 10, 2, 1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 1, lm=0.0, ep=-1e9)
    assert status[0] == 1 and not dx.any()
    assert np.array_equal(poses, s["poses"].numpy())
    poses, disps, dx, dz, status = _run_ba(s, cuda, 1)                           # the workspace is usable afterwards
    assert dx.any()


def test_partitioned_and_one_chain_solves_agree(cuda):
    """the same system through ba_solve_twin_kernel and through the one-chain pipeline (pvo_debug_config, set in a process of its own):
    fp64 rounding apart (the order of elimination differs), far inside fp32"""
    import subprocess
    import sys
    import tempfile
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); from test_geom_ba_gpu import _scene; from pvo_amd import droid_backends as db; "
            "s = _scene(3, 50, 8, 10, 3, 1); d = lambda t: t.cuda(); p, q = d(s['poses'].clone()), d(s['disps'].clone()); "
            "dx, dz = db.ba(p, q, d(s['intr']), d(s['target']), d(s['weight']), d(s['eta']), d(s['ii']), d(s['jj']), s['t0'], s['t1'], 1, 1e-4, 0.1, False); "
            "torch.save((dx.cpu(), dz.cpu(), db.ba_last_partition(s['ii'].shape[0], 49, 50, 80, 'cuda')), sys.argv[1])") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for solver in ("pipe", "twin"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            r = subprocess.run([sys.executable, "-c", _KNOBS + code, f.name, "ba_solver=" + solver], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:]
            outs[solver] = torch.load(f.name)
    assert outs["pipe"][2] == (0, 0) and outs["twin"][2][0] > 0
    scale = outs["pipe"][0].abs().max().item()
    assert (outs["pipe"][0] - outs["twin"][0]).abs().max().item() <= 4e-7 * scale
    assert (outs["pipe"][1] - outs["twin"][1]).abs().max().item() <= 1e-6 * max(1.0, outs["pipe"][1].abs().max().item())


@pytest.mark.parametrize("P,radius", [(40, 6), (70, 9)])
def test_blocked_dense_pose_solve_of_a_densely_connected_global_graph_matches_the_oracle(cuda, P, radius):
    """A global graph connected by proximity rather than as a chain (more than 8 edges per pose, beyond the dense matrix-core solve's
    29 poses: the backend's graph of a real sequence) takes the DENSE factorisation in 48 x 48 blocks over many workgroups
    (dense_panel / dense_update / dense_back kernels; 6 P is and is not a multiple of 48): the oracle's dense fp64 solve, bitwise
    repeatable, no partition reported, and a non-SPD system gives the zero update and a usable workspace afterwards."""
    from pvo_amd import droid_backends as db
    ht, wd = 8, 10
    s = _scene(P * 3 + radius, P, ht, wd, radius, 1)
    assert s["ii"].shape[0] > 8 * (P - 1)
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, P, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert db.ba_last_partition(s["ii"].shape[0], P - 1, P, ht * wd, cuda) == (0, 0)
    assert status[0] == 0 and status[1] == want["K"]
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    assert np.abs(dx - want["dx"]).max() < 1e-4 and np.abs(dz - want["dz"]).max() < 1e-4
    again = _run_ba(s, cuda, 2)
    assert np.array_equal(poses, again[0]) and np.array_equal(disps, again[1])
    bad = _run_ba(s, cuda, 1, lm=0.0, ep=-1e9)
    assert bad[4][0] == 1 and not bad[2].any() and np.array_equal(bad[0], s["poses"].numpy())
    ok = _run_ba(s, cuda, 2)
    assert np.array_equal(poses, ok[0])


def test_ba_non_spd_gives_zero_update(cuda):
    s = _scene(10, 4, 6, 8, 2, 1)
    s["weight"] = torch.zeros_like(s["weight"])
    poses, disps, dx, dz, status = _run_ba(s, cuda, 1, lm=0.0, ep=-1.0)
    assert status[0] == 1 and not dx.any()
    assert np.array_equal(poses, s["poses"].numpy())


def test_ba_converges_full_size_sb(cuda):
    """S-B (BASELINE configs[1]): 8 keyframes, 36 edges, 48x64.  Size-independent property:
    with noise-free targets Gauss-Newton drives the reprojection error to ~0; plus the oracle
    on the same input after 2 iterations (the reference's itrs=2)."""
    from pvo_amd import droid_backends as db
    s = _scene(0, 8, 48, 64, 3, 1, noise=0.0)
    assert s["ii"].shape[0] == 36
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), 1, 8, 2, 1e-4, 0.1)
    poses, disps, dx, dz, status = _run_ba(s, cuda, 2)
    assert np.abs(poses - want["poses"]).max() < 1e-4 and np.abs(disps - want["disps"]).max() < 1e-4
    poses, disps, dx, dz, status = _run_ba(s, cuda, 12)
    intr_all = s["intr"][None].repeat(8, 1)
    c, v = db.reproject(torch.from_numpy(poses).to(cuda), torch.from_numpy(disps).to(cuda), intr_all.to(cuda),
                        s["ii"].to(cuda), s["jj"].to(cuda))
    epe = (c.cpu() - s["target"].permute(0, 2, 3, 1)).norm(dim=-1)
    assert epe.mean() < 5e-3


def test_reprojection_kernels_match_oracle(cuda):
    from pvo_amd import droid_backends as db
    s = _scene(3, 7, 13, 17, 3, 1)
    P = 7
    poses, disps, intr = s["poses_gt"], s["disps_gt"] * (1 + 0.1 * torch.randn(P, 13, 17, generator=torch.Generator().manual_seed(1))), s["intr"]
    poses_n, disps_n, intr_n = poses.numpy(), disps.numpy(), intr.numpy()
    ii, jj = s["ii"], s["jj"]
    d = lambda t: t.to(cuda)
    # frame_distance
    got = db.frame_distance(d(poses), d(disps), d(intr), d(ii), d(jj), 0.3).cpu().numpy()
    assert np.allclose(got, O.frame_distance(poses_n, disps_n, intr_n, ii.numpy(), jj.numpy(), 0.3), rtol=1e-5, atol=1e-5)
    far = poses.clone(); far[1, 2] -= 100.0
    assert db.frame_distance(d(far), d(disps), d(intr), d(torch.tensor([0])), d(torch.tensor([1])), 0.3).item() == 1000.0
    # DepthVideo.distance's bidirectional metric in one launch: bit-identical to two calls and their mean
    d1 = db.frame_distance(d(poses), d(disps), d(intr), d(ii), d(jj), 0.3)
    d2 = db.frame_distance(d(poses), d(disps), d(intr), d(jj), d(ii), 0.3)
    assert torch.equal(db.frame_distance_bidirectional(d(poses), d(disps), d(intr), d(ii), d(jj), 0.3), 0.5 * (d1 + d2))
    # projmap
    c, v = db.projmap(d(poses), d(disps), d(intr), d(ii), d(jj))
    wc, wv = O.projmap(poses_n, disps_n, intr_n, ii.numpy(), jj.numpy())
    assert np.allclose(c.cpu().numpy(), wc, atol=2e-4) and np.array_equal(v.cpu().numpy(), wv)
    # iproj
    assert np.allclose(db.iproj(d(poses), d(disps), d(intr)).cpu().numpy(), O.iproj(poses_n, disps_n, intr_n), rtol=1e-5, atol=1e-5)
    # reproject (python-path semantics, per-frame intrinsics)
    intr_all = intr[None].repeat(P, 1)
    c, v = db.reproject(d(poses), d(disps), d(intr_all), d(ii), d(jj))
    wc, wv = O.reproject(poses_n, disps_n, intr_all.numpy(), ii.numpy(), jj.numpy())
    assert np.allclose(c.cpu().numpy(), wc, atol=2e-4) and np.array_equal(v.cpu().numpy(), wv)
    # depth_filter
    ix = torch.tensor([0, 3, 6]); th = torch.tensor([0.05, 0.1, 0.2])
    got = db.depth_filter(d(poses), d(disps), d(intr), d(ix), d(th)).cpu().numpy()
    want = O.depth_filter(poses_n, disps_n, intr_n, ix.numpy(), th.numpy())
    assert (got != want).mean() < 0.01      # votes flip only where |1/dj - 1/d| sits on the threshold


def test_reproject_matches_reference_python_fixture(cuda):
    from pvo_amd import droid_backends as db
    d = dict(np.load(os.path.join(G, "ba_python_b.npz")))
    P = d["poses"].shape[0]
    t = lambda a: torch.from_numpy(a).to(cuda)
    c, v = db.reproject(t(d["poses"]), t(d["disps"]), t(np.tile(d["intr"][None], (P, 1))), t(d["ii"]), t(d["jj"]))
    assert np.allclose(c.cpu().numpy(), d["reproj_coords"], atol=5e-5)
    assert np.array_equal(v.cpu().numpy(), d["reproj_valid"])


@pytest.mark.gpu
@pytest.mark.parametrize("P,ht,wd,radius", [(8, 24, 32, 3), (22, 12, 16, 3), (40, 10, 12, 2), (64, 8, 12, 3)])
def test_wave_and_blocked_cholesky_are_bit_identical(P, ht, wd, radius):
    """the look-ahead pipeline (the default: wave 0 on the critical path, three worker waves behind it), the one-wave
    (barrier-free) and the four-wave blocked factorisation of the pose system perform the same operations on every entry in
    the same order: poses and depths agree bit for bit (dense-in-LDS and compact-envelope storage, with a loop closure in
    the long windows).  Each solver in its own process: the choice is read once per process."""
    import subprocess
    import sys
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); from test_geom_ba_gpu import _scene; from pvo_amd import droid_backends as db; "
            "s = _scene(3, %d, %d, %d, %d, 1); d = lambda t: t.cuda(); p, q = d(s['poses'].clone()), d(s['disps'].clone()); "
            "db.ba(p, q, d(s['intr']), d(s['target']), d(s['weight']), d(s['eta']), d(s['ii']), d(s['jj']), s['t0'], s['t1'], 2, 1e-4, 0.1, False); "
            "torch.save((p.cpu(), q.cpu()), sys.argv[1])") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), P, ht, wd, radius)
    import tempfile
    outs = []
    for solver in ("wave", "blocked", "pipe"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            r = subprocess.run([sys.executable, "-c", _KNOBS + code, f.name, "ba_solver=" + solver], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:]
            outs.append(torch.load(f.name))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    from test_geom_ba_gpu import _scene as sc
    assert (outs[0][0] - sc(3, P, ht, wd, radius, 1)["poses"]).abs().max() > 1e-5
