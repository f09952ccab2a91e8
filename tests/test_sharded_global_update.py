"""Edge-sharded GLOBAL update (BASELINE config 4: factor-graph edges sharded across GPUs with one pose all-reduce per
Gauss-Newton step).  Two gloo ranks each run FactorGraph.update_lowmem(sharded=ShardedBA) on the edges whose source frame
they own - operator calls, damping and depth updates rank-local, the reduced pose system all-reduced - and must reproduce
the single-process update of the whole graph.  Native steps are answered by the CPU oracle (tests/test_sharded_ba.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(__file__))
from oracle import oracle as O
from pvo_amd.parallel import ShardedBA, shard_edges
from test_sharded_ba import OracleBackend


def _scene():
    from test_geom_ba_gpu import _scene
    return _scene(33, 7, 8, 10, 3, 1)


class _Video:
    def __init__(self, s):
        F, ht, wd = s["disps"].shape
        self.ht, self.wd, self.counter = ht * 8, wd * 8, F
        self.poses, self.disps = s["poses"].clone(), s["disps"].clone()
        self.intrinsics = s["intr"][None].repeat(F, 1).contiguous()
        self.dirty = torch.zeros(F, dtype=torch.bool)
        self.fmaps = torch.zeros(F, ht, wd, 128)
        self.inps = torch.zeros(F, 128, ht, wd)
        self.nets = torch.zeros(F, 128, ht, wd)
        self.segms = torch.zeros(F, 1, ht, wd, dtype=torch.int)
        self.segm_filter, self.thresh = False, 0.5

    def reproject(self, ii, jj):
        c, v = O.reproject(self.poses.numpy(), self.disps.numpy(), self.intrinsics.numpy(), np.asarray(ii), np.asarray(jj))
        return torch.from_numpy(c)[None], torch.from_numpy(v)[None]

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        r = O.ba(self.poses.numpy(), self.disps.numpy(), self.intrinsics[0].numpy(), target.numpy(), weight.numpy(), eta.numpy(),
                 ii.numpy(), jj.numpy(), t0, t1, itrs, lm, ep, motion_only=motion_only)
        self.poses.copy_(torch.from_numpy(r["poses"])); self.disps.copy_(torch.from_numpy(r["disps"]).clamp(min=0.001))


class _Operator:
    """per-edge outputs from tables keyed by the edge (i, j) / the source frame, so that a rank holding a subset of the
    edges sees exactly the values the whole-graph run sees for them"""

    def __init__(self, s):
        g = torch.Generator().manual_seed(7)
        F, ht, wd = s["disps"].shape
        self.key = {(int(i), int(j)): k for k, (i, j) in enumerate(zip(s["ii"], s["jj"]))}
        E = len(self.key)
        self.delta = torch.randn(2, E, ht, wd, 4, generator=g) * 0.3
        self.weight = torch.randn(2, E, ht, wd, 2, generator=g)
        self.delta_m = torch.randn(2, E, ht, wd, 2, generator=g)
        self.damp = torch.rand(2, F, ht, wd, generator=g) * 0.05 + 0.01
        self.step = -1

    def parameters(self):
        return iter(())

    def __call__(self, net, inp, corr, motn, ii, jj, flag=False, **kw):
        idx = torch.tensor([self.key[(int(i), int(j))] for i, j in zip(ii, jj)])
        k = self.step
        return net, self.delta[k][idx][None], self.weight[k][idx][None], self.damp[k][torch.unique(ii)][None], {}, self.delta_m[k][idx][None]


def _run(s, ii, jj, sharded, steps=2):
    import pvo_amd.modules.corr as corr_mod
    from pvo_amd.factor_graph import FactorGraph
    F, ht, wd = s["disps"].shape

    class FakeAlt:
        def __init__(self, fmaps, *a, **k):
            pass

        def __call__(self, coords, ii_, jj_):
            return torch.zeros(1, ii_.shape[0], 196, ht, wd)
    corr_mod.AltCorrBlock = FakeAlt
    v, op = _Video(s), _Operator(s)
    # count reprojections to know which global step the operator is in (one reproject per step)
    real_reproject = v.reproject

    def reproject(a, b):
        op.step += 1
        return real_reproject(a, b)
    v.reproject = reproject
    fg = FactorGraph(v, op, device="cpu", corr_impl="alt")
    op.step = -2                                               # add_factors reprojects once
    fg.add_factors(list(ii), list(jj))
    op.step = -1
    fg.update_lowmem(steps=steps, sharded=sharded)
    return v


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = _scene()
    ii, jj, _ = shard_edges(s["ii"].tolist(), s["jj"].tolist(), world, rank)
    sb = ShardedBA(backend=OracleBackend())
    before = s["disps"].clone()
    v = _run(s, ii, jj, sb)
    sb.sync_disps(v.disps, before)
    out[rank] = (v.poses.numpy().copy(), v.disps.numpy().copy(), len(ii))
    dist.destroy_process_group()


def test_two_rank_sharded_global_update_equals_single_process():
    s = _scene()
    whole = _run(s, s["ii"].tolist(), s["jj"].tolist(), None)                          # DepthVideo.ba on the whole graph
    one = _run(s, s["ii"].tolist(), s["jj"].tolist(), ShardedBA(backend=OracleBackend()))   # sharded code path, one rank
    assert np.abs(one.poses.numpy() - whole.poses.numpy()).max() < 2e-5
    assert np.abs(one.disps.numpy() - whole.disps.numpy()).max() < 2e-5
    assert np.abs(whole.poses.numpy() - s["poses"].numpy()).max() > 1e-4               # the update did move the poses
    world = 2
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29531, out), nprocs=world, join=True)
    assert out[0][2] + out[1][2] == s["ii"].shape[0] and min(out[0][2], out[1][2]) > 0
    for r in range(world):
        assert np.abs(out[r][0] - whole.poses.numpy()).max() < 5e-5
        assert np.abs(out[r][1] - whole.disps.numpy()).max() < 5e-5
    assert np.array_equal(out[0][0], out[1][0])                                        # pose replicas bit-identical


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("nkf,H,W,closure", [(24, 16, 24, True), (64, 48, 64, False)])
def test_hip_sharded_global_update_two_virtual_ranks_bit_identical_to_whole_graph(cuda, nkf, H, W, closure):
    """The NATIVE path (resident volumes, pvo_graph_update per step, pvo_ba_local / _finish around the all-reduce of the
    envelope) for a 24-keyframe global update: two virtual ranks in ONE process - two threads that take turns on the device
    (a lock around every stretch of GPU work, so that no kernel of one rank is resident beside a kernel of the other; two
    processes sharing one GPU is exactly the co-residency DESIGN.md section 5 shows to be unsafe for the BA) and exchange the
    pose system through memory - must give the poses of the whole graph on one GPU BIT FOR BIT, and its depth maps after the
    per-rank updates are merged."""
    # (64, 48, 64): BASELINE.json configs[3] (S-20) at its full size - 64 keyframes, 372 edges, 48x64 maps
    import threading
    import bench
    from pvo_amd.parallel import ShardedBA, shard_edges
    steps = 2
    ii = [i for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 3] + ([2, 20] if closure else [])
    jj = [j for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 3] + ([20, 2] if closure else [])  # + a loop closure
    assert nkf != 64 or len(ii) == 372
    buf = nkf + 8
    intr = (15.0, 15.0, 12.0, 8.0) if H == 16 else (40.0, 40.0, 32.0, 24.0)
    g = torch.Generator().manual_seed(3)
    noise = 0.5 * torch.randn(len(ii), H, W, 2, generator=g)
    wts = torch.rand(len(ii), H, W, 2, generator=g)
    pos = {e: k for k, e in enumerate(zip(ii, jj))}
    turn = threading.Lock()

    def build(ii_l, jj_l):
        video, graph = bench.make_window(cuda, seed=7, H8=H, W8=W, NKF=nkf, buffer=buf, corr_impl="volume", add_edges=False, max_factors=-1,
                                         intr=intr)
        video.counter = nkf
        graph.add_factors(ii_l, jj_l)
        sel = torch.tensor([pos[e] for e in zip(graph._ii_h, graph._jj_h)])
        graph.target_cam = graph.target_cam + noise[sel].to(cuda)[None]
        graph.weight = wts[sel].to(cuda)[None].contiguous()
        torch.cuda.synchronize()
        return video, graph

    # the whole graph on one GPU (ShardedBA without communication: the same code path)
    video, graph = build(ii, jj)
    d0 = video.disps.clone()
    graph.update_lowmem(steps=steps, sharded=ShardedBA(structure=(ii, jj), communicate=False))
    torch.cuda.synchronize()
    whole_p, whole_d = video.poses.clone(), video.disps.clone()
    assert (whole_p[:nkf] - bench.make_window(cuda, seed=7, H8=H, W8=W, NKF=nkf, buffer=buf, add_edges=False)[0].poses[:nkf]).abs().max() > 1e-4

    world = 2
    box, meet = [None] * world, threading.Barrier(world)

    class Virtual(ShardedBA):
        def __init__(self, rank):
            super().__init__(structure=(ii, jj))
            self.rank = rank

        def _world(self):
            return world

        def _allreduce(self, t):
            torch.cuda.synchronize()
            box[self.rank] = t.clone()
            turn.release()                                    # hand the device over while waiting for the other rank
            meet.wait()
            total = box[0] + box[1]
            meet.wait()
            turn.acquire()
            t.copy_(total)

    out, errs = [None] * world, []

    def rank_main(r):
        try:
            ii_l, jj_l, _ = shard_edges(ii, jj, world, r)
            with turn:
                v, gph = build(ii_l, jj_l)
            turn.acquire()
            try:
                gph.update_lowmem(steps=steps, sharded=Virtual(r))
                torch.cuda.synchronize()
            finally:
                turn.release()
            out[r] = (v.poses.clone(), v.disps.clone(), len(ii_l))
        except Exception as e:                                # noqa: BLE001
            errs.append(repr(e))
            meet.abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert out[0][2] + out[1][2] == len(ii) and min(out[0][2], out[1][2]) > 0
    assert torch.equal(out[0][0], out[1][0])                  # replicas agree ...
    assert torch.equal(out[0][0], whole_p)                    # ... with the whole graph, bit for bit
    merged = d0 + (out[0][1] - d0) + (out[1][1] - d0)
    assert (merged - whole_d).abs().max() < 1e-6


@pytest.mark.gpu
def test_hip_sharded_ba_through_a_one_rank_rccl_group(cuda):
    """The collective of the edge-sharded BA on the path an 8-GPU run takes - `dist.all_reduce` of the int64 ENVELOPE message on an
    RCCL ("nccl") process group, between pvo_ba_local and pvo_ba_finish - executed with the one rank a 1-GPU box has: the sum over
    one rank is the identity, so poses and depths must equal, bit for bit, the same BA without communication; and the message
    must be the envelope (not the dense system)."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(__file__))
    from test_geom_ba_gpu import _scene
    from pvo_amd.parallel import ShardedBA, envelope_structure
    P, ht, wd = 24, 16, 24
    s = _scene(31, P, ht, wd, 3, 1)
    args = lambda: (s["poses"].clone().to(cuda), s["disps"].clone().to(cuda))
    dev = lambda k: s[k].to(cuda)
    calls = []

    class Counting(ShardedBA):
        def _allreduce(self, t):
            calls.append((t.dtype, t.numel(), t.device.type))
            super()._allreduce(t)

    def run(sb):
        poses, disps = args()
        sb.ba(poses, disps, dev("intr"), dev("target"), dev("weight"), dev("eta"), dev("ii"), dev("jj"), 1, P, itrs=2, lm=1e-4, ep=0.1)
        torch.cuda.synchronize()
        return poses.cpu(), disps.cpu()
    structure = (s["ii"].tolist(), s["jj"].tolist())
    plain = run(ShardedBA(structure=structure, communicate=False))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29541"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        assert dist.get_backend() == "nccl"
        sb = Counting(structure=structure)
        sb.always_pack, sb.collective_at_one = True, True
        got = run(sb)
    finally:
        dist.destroy_process_group()
    assert len(calls) == 2                                              # one collective per Gauss-Newton step
    first = envelope_structure(structure[0], structure[1], 1, P)
    n_env = sum(36 * (b - first[b] + 1) for b in range(P - 1)) + 6 * (P - 1)
    assert all(c == (torch.int64, n_env, "cuda") for c in calls), (calls, n_env)
    assert n_env < (6 * (P - 1)) ** 2 // 2
    assert torch.equal(got[0], plain[0]) and torch.equal(got[1], plain[1])
    assert (plain[0] - s["poses"]).abs().max() > 1e-4                 # the BA did move the poses
