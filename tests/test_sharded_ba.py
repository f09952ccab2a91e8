"""Edge-sharded BA.
CPU: two gloo ranks run pvo_amd.parallel.ShardedBA over an ORACLE-backed native layer (the partition,
eta-row selection and the single all-reduce are the code under test); the result must equal the
single-process oracle on the whole graph.
GPU: the HIP split entry points (plan/local/finish) with two shards in one process, summed by hand,
against pvo_ba on the whole graph."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(__file__))
from oracle import oracle as O
from pvo_amd.parallel import ShardedBA, envelope_index, envelope_structure, local_eta_rows, partition_by_source


class OracleBackend:
    """the three native steps, computed by the CPU oracle on torch CPU tensors"""

    def ba_workspace(self, E, P, F, HW, device):
        return {}

    def ba_plan(self, ii, jj, F, HW, K_eta, t0, t1, ws):
        ws.clear()

    def ba_local(self, poses, disps, intr, targets, weights, eta, ii, jj, t0, t1, motion_only, sys_buf, ws):
        r = O.ba(poses.numpy(), disps.numpy(), intr.numpy(), targets.numpy(), weights.numpy(),
                 None if eta is None else eta.numpy(), ii.numpy(), jj.numpy(), t0, t1, 1, 0.0, 1.0,
                 motion_only=motion_only, want_sys=True)
        sys_buf.copy_(torch.from_numpy(r["sys"]))
        ws["args"] = (intr, targets, weights, eta)

    def ba_finish(self, poses, disps, sys_buf, ii, jj, t0, t1, lm, ep, motion_only, ws):
        # solve the reduced system exactly as oracle_ba does, then let the oracle redo this rank's
        # iteration with dx forced to the global solution: depth back-substitution is rank-local
        n = 6 * (t1 - t0)
        A = sys_buf[:n * n].view(n, n).numpy().copy(); b = sys_buf[n * n:].numpy().copy()
        A = np.tril(A) + np.tril(A, -1).T              # a Cholesky reads the lower triangle only (what the envelope all-reduce completes)
        A[np.diag_indices(n)] += ep + lm * np.diag(A)
        dx = np.linalg.solve(A, b).astype(np.float32).reshape(-1, 6)
        intr, targets, weights, eta = ws["args"]
        r = O.ba_apply(poses.numpy(), disps.numpy(), intr.numpy(), targets.numpy(), weights.numpy(),
                       None if eta is None else eta.numpy(), ii.numpy(), jj.numpy(), t0, t1, dx, motion_only)
        poses.copy_(torch.from_numpy(r["poses"])); disps.copy_(torch.from_numpy(r["disps"]))
        return [torch.from_numpy(dx), None]


def _graph():
    from test_geom_ba_gpu import _scene
    return _scene(21, 6, 8, 10, 2, 1)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = _graph()
    owner, _ = partition_by_source(s["ii"].tolist(), world)
    mine = torch.tensor([o == rank for o in owner])
    ii, jj = s["ii"][mine], s["jj"][mine]
    rows = local_eta_rows(s["ii"].tolist(), ii.tolist(), s["t0"], s["t1"])
    poses, disps = s["poses"].clone(), s["disps"].clone()
    before = disps.clone()
    sb = ShardedBA(backend=OracleBackend())
    sb.ba(poses, disps, s["intr"], s["target"][mine].contiguous(), s["weight"][mine].contiguous(),
          s["eta"][rows].contiguous(), ii.contiguous(), jj.contiguous(), s["t0"], s["t1"], itrs=2)
    sb.sync_disps(disps, before)
    # the same with only the ENVELOPE of the pose system all-reduced (structure from the global edge list)
    poses2, disps2 = s["poses"].clone(), s["disps"].clone()
    sb2 = ShardedBA(backend=OracleBackend(), structure=(s["ii"].tolist(), s["jj"].tolist()))
    sb2.ba(poses2, disps2, s["intr"], s["target"][mine].contiguous(), s["weight"][mine].contiguous(),
           s["eta"][rows].contiguous(), ii.contiguous(), jj.contiguous(), s["t0"], s["t1"], itrs=2)
    sb2.sync_disps(disps2, before)
    assert 0 < sb2.last_message_bytes <= sb.last_message_bytes
    assert torch.equal(poses2, poses) and torch.equal(disps2, disps)      # bit-identical to the dense all-reduce
    out[rank] = (poses.numpy().copy(), disps.numpy().copy())
    dist.destroy_process_group()


def test_partition_is_deterministic_and_balanced():
    ii = [0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 4]
    owner, by_frame = partition_by_source(ii, 2)
    assert owner == partition_by_source(ii, 2)[0]
    assert all(owner[k] == by_frame[f] for k, f in enumerate(ii))          # all edges of a source frame together
    loads = [owner.count(r) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= 2
    assert local_eta_rows([0, 1, 2, 3], [0, 2], 1, 4) == [0, 1, 2, 3]      # window frames are always present
    assert local_eta_rows([0, 5, 6], [6], 1, 4) == [1, 2, 3, 5]


def test_envelope_structure_covers_every_coupling_of_the_pose_system():
    """first[b] from the edge list must bound the numeric envelope of the assembled + Schur-reduced system (oracle, dense):
    window graphs, long chains with loop closures, fixed poses below t0"""
    from test_geom_ba_gpu import _scene
    rng = np.random.default_rng(0)
    for seed, P, radius, t0, extra in ((1, 6, 2, 1, []), (2, 9, 3, 2, [(1, 7), (7, 1)]), (3, 12, 1, 1, [(0, 11), (11, 0), (3, 9)])):
        s = _scene(seed, P, 6, 8, radius, t0)
        ii = s["ii"].tolist() + [e[0] for e in extra]; jj = s["jj"].tolist() + [e[1] for e in extra]
        E = len(ii)
        tgt = torch.cat([s["target"], s["target"][:len(extra)]]); wgt = torch.cat([s["weight"], s["weight"][:len(extra)]])
        r = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), tgt.numpy(), wgt.numpy(), s["eta"].numpy(),
                 np.asarray(ii), np.asarray(jj), s["t0"], s["t1"], 1, 0.0, 1.0, want_sys=True)
        n = 6 * (s["t1"] - s["t0"])
        A = r["sys"][:n * n].reshape(n, n)
        first = envelope_structure(ii, jj, s["t0"], s["t1"])
        assert len(first) == n // 6 and all(0 <= f <= b for b, f in enumerate(first))
        blk = np.abs(A).reshape(n // 6, 6, n // 6, 6).max(axis=(1, 3)) > 0
        for b in range(n // 6):
            cols = np.nonzero(blk[b, :b + 1])[0]
            assert cols.min() >= first[b], (seed, b, cols.min(), first[b])
        idx = envelope_index(first, "cpu")
        assert idx.numel() == 36 * sum(b - f + 1 for b, f in enumerate(first)) + n and idx.unique().numel() == idx.numel()
        lower = np.tril(A)
        keep = np.zeros(n * n + n, bool); keep[idx.numpy()] = True
        assert not lower.reshape(-1)[~keep[:n * n]].any()                 # nothing of the lower triangle is left out


def test_packed_message_length_of_the_library_equals_the_index_path():
    """pvo_ba_packed_elems (host helper of the C ABI; no GPU needed) counts the entries envelope_index enumerates: the library's
    packed message and the index_select message are the same entries (the GPU test compares their sums bit for bit)"""
    from pvo_amd import droid_backends as db
    g = np.random.default_rng(5)
    for P in (1, 7, 21, 63, 200):
        first = [int(g.integers(0, b + 1)) if g.random() < 0.7 else b for b in range(P)]
        assert db.ba_packed_elems(first) == envelope_index(first, "cpu").numel() == 36 * sum(b - f + 1 for b, f in enumerate(first)) + 6 * P
    assert db.ba_packed_elems([0, 5, 9]) == db.ba_packed_elems([0, 1, 2])          # entries beyond the diagonal are clamped to it


def test_two_rank_sharded_ba_equals_single_process_gloo():
    world = 2
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, out), nprocs=world, join=True)
    s = _graph()
    want = O.ba(s["poses"].numpy(), s["disps"].numpy(), s["intr"].numpy(), s["target"].numpy(), s["weight"].numpy(),
                s["eta"].numpy(), s["ii"].numpy(), s["jj"].numpy(), s["t0"], s["t1"], 2, 1e-4, 0.1)
    for r in range(world):
        poses, disps = out[r]
        assert np.abs(poses - want["poses"]).max() < 2e-5
        assert np.abs(disps - want["disps"]).max() < 2e-5
    assert np.array_equal(out[0][0], out[1][0])                           # pose replicas bit-identical


@pytest.mark.gpu
def test_hip_split_entry_points_two_shards_equal_whole_graph(cuda):
    from pvo_amd import droid_backends as db
    from test_geom_ba_gpu import _scene
    s = _scene(5, 8, 24, 32, 3, 1)
    d = lambda t: t.to(cuda)
    poses_w, disps_w = d(s["poses"].clone()), d(s["disps"].clone())
    db.ba(poses_w, disps_w, d(s["intr"]), d(s["target"]), d(s["weight"]), d(s["eta"]), d(s["ii"]), d(s["jj"]),
          s["t0"], s["t1"], 2, 1e-4, 0.1, False)
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
        sh["sys"] = torch.zeros((6 * P) ** 2 + 6 * P, dtype=torch.int64, device=cuda)      # fixed point: integer sums
        db.ba_plan(sh["ii"], sh["jj"], F, ht * wd, sh["eta"].shape[0], s["t0"], s["t1"], sh["ws"])
        shards.append(sh)
    for _ in range(2):
        for sh in shards:
            db.ba_local(sh["poses"], sh["disps"], d(s["intr"]), sh["target"], sh["weight"], sh["eta"], sh["ii"], sh["jj"],
                        s["t0"], s["t1"], False, sh["sys"], sh["ws"])
        total = shards[0]["sys"] + shards[1]["sys"]                        # what the RCCL all-reduce produces
        for sh in shards:
            db.ba_finish(sh["poses"], sh["disps"], total.clone(), sh["ii"], sh["jj"], s["t0"], s["t1"], 1e-4, 0.1, False, sh["ws"])
            sh["sys"].zero_()                                                # (finish zeroes the buffer IT was given)
    assert torch.equal(shards[0]["poses"], shards[1]["poses"])            # replicas bit-identical
    # ... and, the system being an integer sum, bit-identical to the whole graph on one GPU
    assert torch.equal(shards[0]["poses"], poses_w)
    merged = s["disps"].clone().to(cuda) + sum(sh["disps"] - d(s["disps"]) for sh in shards)
    assert (merged - disps_w).abs().max() < 2e-6


@pytest.mark.gpu
def test_ba_is_bitwise_reproducible(cuda):
    """the pose system is accumulated with integer (fixed-point) atomics: repeated runs give identical bits"""
    from pvo_amd import droid_backends as db
    from test_geom_ba_gpu import _scene
    s = _scene(7, 8, 48, 64, 3, 1)
    d = lambda t: t.to(cuda)
    outs = []
    for _ in range(4):
        poses, disps = d(s["poses"].clone()), d(s["disps"].clone())
        db.ba(poses, disps, d(s["intr"]), d(s["target"]), d(s["weight"]), d(s["eta"]), d(s["ii"]), d(s["jj"]),
              s["t0"], s["t1"], 2, 1e-4, 0.1, False)
        outs.append((poses.clone(), disps.clone()))
    for p, q in outs[1:]:
        assert torch.equal(p, outs[0][0]) and torch.equal(q, outs[0][1])


@pytest.mark.gpu
@pytest.mark.parametrize("nf,ht,wd", [(64, 16, 24), (30, 8, 10), (8, 12, 16)])
def test_envelope_allreduce_path_is_bit_identical_on_the_hip_solver(cuda, nf, ht, wd):
    """ShardedBA with the envelope message forced on one rank - packed by the library (pvo_ba_pack / pvo_ba_finish_packed: the
    message is read in place of the dense system) and, as before round 4, by index_select / index_copy_ around the dense
    system - against the dense message: poses and depths bit-identical.  64 keyframes: the partitioned solve; 8: a window-sized
    system, which the packed path factorises from the compact image too."""
    from test_geom_ba_gpu import _scene
    s = _scene(13, nf, ht, wd, 3, 1)
    d = lambda t: t.to(cuda)
    outs = []
    for structure, torch_pack in ((None, False), ((s["ii"].tolist(), s["jj"].tolist()), False), ((s["ii"].tolist(), s["jj"].tolist()), True)):
        poses, disps = d(s["poses"].clone()), d(s["disps"].clone())
        sb = ShardedBA(structure=structure)
        sb.always_pack, sb.torch_pack = True, torch_pack
        sb.ba(poses, disps, d(s["intr"]), d(s["target"]), d(s["weight"]), d(s["eta"]), d(s["ii"]), d(s["jj"]), s["t0"], s["t1"], itrs=2,
              lm=1e-5, ep=1e-2)
        outs.append((poses.clone(), disps.clone(), sb.last_message_bytes))
    for k in (1, 2):
        assert torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1]), k
    assert (outs[0][0] - d(s["poses"])).abs().max() > 1e-4
    n6 = 6 * (s["t1"] - s["t0"])
    assert outs[1][2] == outs[2][2]                                       # the same entries either way
    if nf == 64:
        assert 0 < outs[1][2] < 0.2 * 8 * (n6 * n6 + n6)                  # 63 free poses, radius 3: ~12 % of the dense message


@pytest.mark.gpu
@pytest.mark.parametrize("nf,ht,wd,dt", [(8, 48, 64, torch.float16), (30, 12, 16, torch.float16), (8, 24, 32, torch.bfloat16)])
def test_pose_solve_with_a_convolution_riding_in_its_dispatch(cuda, nf, ht, wd, dt):
    """pvo_ba_finish_conv1x1: the one-workgroup pose solve and an independent 1x1 convolution (GraphAgg's upsampling mask in
    pvo_graph_update) share a dispatch.  Poses and depths are bit-identical to the plain call over repeated runs, and the
    convolution equals pvo_conv1x1_c128 bit for bit (window-sized system in LDS, and a 29-pose system on the envelope path)."""
    from pvo_amd import droid_backends as db
    from test_geom_ba_gpu import _scene
    s = _scene(11, nf, ht, wd, 3, 1)
    d = lambda t: t.to(cuda)
    F = s["disps"].shape[0]
    P = s["t1"] - s["t0"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 128, 48, 64, generator=g).to(dt).to(cuda).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(576, 128, generator=g) * 0.05).to(dt).to(cuda)
    b = torch.randn(576, generator=g).to(cuda)
    want_y = db.conv1x1_c128(x, w, b)

    def run(rider):
        poses, disps = d(s["poses"].clone()), d(s["disps"].clone())
        ws = db.ba_workspace(s["ii"].shape[0], P, F, ht * wd, cuda)
        sys_buf = torch.zeros((6 * P) ** 2 + 6 * P, dtype=torch.int64, device=cuda)
        db.ba_plan(d(s["ii"]), d(s["jj"]), F, ht * wd, s["eta"].shape[0], s["t0"], s["t1"], ws)
        ys = []
        for it in range(2):
            db.ba_local(poses, disps, d(s["intr"]), d(s["target"]), d(s["weight"]), d(s["eta"]), d(s["ii"]), d(s["jj"]),
                        s["t0"], s["t1"], False, sys_buf, ws, sys_is_zero=it > 0)
            out = db.ba_finish(poses, disps, sys_buf, d(s["ii"]), d(s["jj"]), s["t0"], s["t1"], 1e-4, 0.1, False, ws,
                               outputs=False, rider=(x, w, b) if rider and it == 0 else None)
            if len(out) == 3:
                ys.append(out[2])
        return poses, disps, ys

    p0, q0, _ = run(False)
    assert (p0.cpu() - s["poses"]).abs().max() > 1e-5
    for _ in range(6):
        p1, q1, ys = run(True)
        assert torch.equal(p0, p1) and torch.equal(q0, q1)
        assert len(ys) == 1 and torch.equal(ys[0].view(torch.int16), want_y.view(torch.int16))
