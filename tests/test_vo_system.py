"""The VO system around the hot path: motion filter, frontend, global-BA backend (alt-corr + update_lowmem),
trajectory filler, Droid.  Two kinds of test:
  * closed loop on a synthetic plane scene with ground-truth correspondences in place of the learned operator:
    the backend must not degrade the frontend's trajectory and the filler must place the non-keyframes;
  * the real (randomly initialised) network end to end: every code path runs on the HIP kernels and produces a
    finite, well-formed trajectory (no checkpoint or dataset exists in this environment, so accuracy with
    learned weights cannot be measured here)."""
from argparse import Namespace

import os

import numpy as np
import pytest
import torch

from pvo_amd.synthetic import OracleFlowOperator, PlaneScene, run_sequence
from pvo_amd.trajectory import ate_rmse, camera_centres


class _TimeStampedOperator(OracleFlowOperator):
    """frames without a binding (the filler's temporary slots) are identified by their time stamp"""

    def _gt_tables(self, nslots):
        ts = self.video.tstamp[:nslots].round().long().tolist()
        for s in range(nslots):
            f = self.frame_of.get(s, ts[s])
            self.gt_poses[s] = self.scene.poses[f]; self.gt_disps[s] = self.scene.disps[f]
        return self.gt_poses.to(self.dev), self.gt_disps.to(self.dev)


class _RandomFeatures(torch.nn.Module):
    def forward(self, x):
        b, n, _, h, w = x.shape
        g = torch.Generator().manual_seed(int(x.shape[1]))
        return torch.randn(b, n, 128, h // 8, w // 8, generator=g).to(x.device)


def test_depth_video_item_access_and_normalize():
    from pvo_amd.depth_video import DepthVideo
    v = DepthVideo(image_size=(32, 48), buffer=8, device="cpu")
    pose = torch.tensor([1.0, 2.0, 3.0, 0, 0, 0, 1.0])
    v[0] = (0.0, None, pose, 2.0, torch.tensor([10.0, 10.0, 3.0, 2.0]), torch.zeros(128, 4, 6), torch.zeros(128, 4, 6),
            torch.ones(128, 4, 6))
    v[1] = (1.0, None, pose * 2, 4.0, None)
    assert v.counter == 2
    p, d, k, f, n, i = v[-1]
    assert torch.equal(p, pose * 2) and float(d.mean()) == 4.0 and f.shape == (4, 6, 128)
    v[2:4] = (torch.tensor([2.0, 3.0]), None, torch.stack([pose, pose]), 1.0, None, torch.zeros(2, 128, 4, 6))
    assert v.counter == 2                       # slices do not move the counter (the filler does that itself)
    v.normalize()
    assert abs(float(v.disps[:2].mean()) - 1.0) < 1e-6
    assert torch.allclose(v.poses[0, :3], pose[:3] * 3.0)      # mean disparity was 3
    assert bool(v.dirty[:2].all()) and not bool(v.dirty[2])
    v.upsample(torch.tensor([0]), torch.zeros(1, 576, 4, 6))
    assert v.disps_up.shape == (8, 32, 48)


@pytest.mark.gpu
def test_backend_and_filler_closed_loop(cuda):
    from pvo_amd import droid_backends as db
    from pvo_amd.backend import DroidBackend
    from pvo_amd.depth_video import DepthVideo
    from pvo_amd.frontend import DroidFrontend
    from pvo_amd.trajectory_filler import PoseTrajectoryFiller
    scene = PlaneScene(ht=24, wd=32, n_frames=14, seed=0)
    video = DepthVideo(image_size=(scene.ht * 8, scene.wd * 8), buffer=64, device=cuda)
    op = _TimeStampedOperator(scene, video, lambda p, d, k, i, j: db.reproject(p, d, k, i, j)[0])
    fe = DroidFrontend(op, video, device=cuda, warmup=8, keyframe_thresh=0.5, frontend_thresh=16.0, frontend_window=20,
                       frontend_radius=2, frontend_nms=1)
    poses_fe, frames = run_sequence(scene, video, fe, op)
    # the reference keeps a dropped keyframe's time stamp on the slot (rm_keyframe does not move tstamp); the
    # filler needs true stamps here, so set them from the bindings
    for s, f in enumerate(frames):
        video.tstamp[s] = float(f)
    gt = camera_centres(scene.poses[frames].numpy())
    ate_fe = ate_rmse(camera_centres(poses_fe.numpy()), gt)

    net = Namespace(cnet=None, fnet=_RandomFeatures(), update=op)
    args = Namespace(device=cuda, beta=0.3, backend_thresh=16.0, backend_radius=2, backend_nms=2)
    DroidBackend(net, video, args)(steps=4)
    n = video.counter
    poses_be = video.poses[:n].cpu()
    ate_be = ate_rmse(camera_centres(poses_be.numpy()), gt)
    print("ATE frontend %.5f -> after global BA %.5f" % (ate_fe, ate_be))
    assert torch.isfinite(poses_be).all() and abs(float(video.disps[:n].mean()) - 1.0) < 0.2
    assert ate_be < max(2 * ate_fe, 0.02 * np.linalg.norm(gt[-1] - gt[0]))

    # fill every scene frame (keyframes and the frames the frontend dropped alike)
    stream = [(float(k), torch.zeros(3, scene.ht * 8, scene.wd * 8), scene.intr * 8.0, None) for k in range(scene.n)]
    traj = PoseTrajectoryFiller(net, video, cuda)(stream)
    assert video.counter == n                                  # temporary slots were released
    est = traj.data.cpu()
    assert est.shape == (scene.n, 7) and torch.isfinite(est).all()
    ate_all = ate_rmse(camera_centres(est.numpy()), camera_centres(scene.poses.numpy()))
    print("ATE over all %d frames after filling: %.5f" % (scene.n, ate_all))
    assert ate_all < 0.05 * np.linalg.norm(gt[-1] - gt[0])


def _textured_stream(n, ht, wd, seed=0):
    """a drifting random texture: enough apparent motion for the filter, content is irrelevant"""
    g = torch.Generator().manual_seed(seed)
    big = torch.randint(0, 256, (3, ht + 64, wd + 8 * n + 64), generator=g).float()
    big = torch.nn.functional.avg_pool2d(big[None], 5, stride=1, padding=2)[0]
    intr = torch.tensor([wd * 0.8, wd * 0.8, wd / 2.0, ht / 2.0])
    for t in range(n):
        yield t, big[:, 16:16 + ht, 8 * t:8 * t + wd].contiguous(), intr.clone(), None


@pytest.mark.gpu
def test_droid_end_to_end_with_random_weights(cuda):
    from pvo_amd.droid import Droid, default_args
    ht, wd, n = 128, 160, 14
    torch.manual_seed(0)
    args = default_args(device="cuda:0", image_size=[ht, wd], buffer=64, warmup=8, filter_thresh=0.0,
                        keyframe_thresh=0.0, frontend_thresh=100.0, backend_thresh=100.0)
    droid = Droid(args)
    for t, image, intr, segm in _textured_stream(n, ht, wd):
        droid.track(t, image, intrinsics=intr, segments=segm)
    assert droid.video.counter == n                            # filter_thresh 0: every frame is a keyframe
    assert droid.frontend.is_initialized and len(droid.frontend.graph._ii_h) > 0
    kf = droid.get_traj()
    assert kf.shape == (n, 7) and np.isfinite(kf).all()
    assert droid.get_depth().shape == (n, ht, wd) and droid.get_flow().shape == (1, n, ht, wd, 2)
    traj = droid.terminate(_textured_stream(n, ht, wd), need_inv=True)
    assert traj.shape == (n, 7) and np.isfinite(traj).all()
    assert np.allclose(np.linalg.norm(traj[:, 3:], axis=1), 1.0, atol=1e-3)
    assert float(droid.video.disps[:n].min()) >= 0.001


def test_handoff_files_round_trip(tmp_path):
    from pvo_amd.handoff import save_flow_depth, write_kitti_trajectory
    flow, disp = torch.randn(16, 24, 2), torch.rand(16, 24)
    p_flow, p_depth = save_flow_depth(str(tmp_path), "0001_00012", flow, disp, valid=torch.ones(16, 24, 1))
    assert p_flow.endswith("full_flow/0001_00012.npy") and p_depth.endswith("depth/0001_00012.npy")
    assert np.array_equal(np.load(p_flow), flow.numpy()) and np.array_equal(np.load(p_depth), disp.numpy())
    p2, _ = save_flow_depth(str(tmp_path), "r", flow, disp, resize_hw=(32, 48))
    assert np.load(p2).shape == (32, 48, 2)
    traj = np.array([[1.0, 2.0, 3.0, 0, 0, 0, 1.0], [0, 0, 0, 0, 0, np.sin(0.25), np.cos(0.25)]])
    write_kitti_trajectory(str(tmp_path / "traj.txt"), traj)
    rows = np.loadtxt(str(tmp_path / "traj.txt"))
    assert rows.shape == (2, 12) and np.allclose(rows[0], [1, 0, 0, 1, 0, 1, 0, 2, 0, 0, 1, 3])
    assert np.allclose(rows[1, [0, 1, 4, 5]], [np.cos(0.5), -np.sin(0.5), np.sin(0.5), np.cos(0.5)], atol=1e-9)


def test_sequence_reader_and_pose_file_parser(tmp_path):
    """tools/test_vo.py: VKITTI2 directory layout -> (t, BGR image, scaled intrinsics, 1/8 segment ids); extrinsic.txt
    -> camera-to-world poses (every second row, inverted)"""
    import importlib.util
    import os
    from PIL import Image
    spec = importlib.util.spec_from_file_location("test_vo_tool", os.path.join(os.path.dirname(__file__), "..", "tools", "test_vo.py"))
    tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
    root = tmp_path / "Scene01"
    (root / "15-deg-left" / "frames" / "rgb" / "Camera_0").mkdir(parents=True)
    (root / "15-deg-left" / "panFPN_segm").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for k in range(3):
        Image.fromarray(rng.integers(0, 255, (60, 200, 3), dtype=np.uint8)).save(root / "15-deg-left" / "frames" / "rgb" / "Camera_0" / ("rgb_%05d.jpg" % k))
        seg = np.zeros((60, 200, 3), dtype=np.uint8); seg[:, 100:, 0] = 7; seg[:, 100:, 1] = 1      # id 7 + 256
        Image.fromarray(seg).save(root / "15-deg-left" / "panFPN_segm" / ("seg_%05d.png" % k))
    frames = list(tool.image_stream(str(root), image_size=(30, 101), mode="val", segm_filter=True))
    assert len(frames) == 3
    t, image, intr, segm = frames[1]
    assert t == 1 and image.shape == (3, 24, 96) and image.dtype == torch.int32 and segm.shape == (1, 1, 3, 12)
    assert torch.allclose(intr, torch.tensor([725.0087 * 101 / 200, 725.0087 * 101 / 200, 620.5 * 30 / 60, 187.0 * 30 / 60]))
    assert set(segm.unique().tolist()) <= {0, 263}
    assert tool.rgb2id(np.array([[[1, 2, 3]]]))[0, 0] == 1 + 512 + 3 * 65536
    # pose file: header + two cameras per frame
    T = np.eye(4); T[:3, 3] = [1.0, 2.0, 3.0]
    rows = ["frame cameraID r11 ..."]
    for f in range(2):
        for cam in range(2):
            M = T.copy(); M[0, 3] += f + 10 * cam
            rows.append("%d %d " % (f, cam) + " ".join("%.6f" % v for v in M.reshape(-1)))
    (root / "15-deg-left" / "extrinsic.txt").write_text("\n".join(rows) + "\n")
    poses = tool.read_vkitti2_poses(str(root / "15-deg-left" / "extrinsic.txt"))
    assert poses.shape == (2, 4, 4) and np.allclose(poses[1][:3, 3], [-2.0, -2.0, -3.0])
    assert tool.parse_args(["--datapath", "x"]).frontend_window == 25


def test_trajectory_filler_matches_reference(monkeypatch):
    """PoseTrajectoryFiller._fill against the reference's own __fill (tests/golden/gen_golden.py: gen_filler): bracketing
    keyframes, se(3) interpolation, edges to both bracketing keyframes, six motion-only updates over the temporary slots"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from gen_golden import filler_case
    import pvo_amd.trajectory_filler as tf
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "trajectory_filler.npz"))
    ts, poses, stamps = filler_case()
    N, M, ht, wd = ts.shape[0], len(stamps), 16, 24
    rec = {"calls": []}

    class Video:
        def __setitem__(self, index, item):
            rec["set_index"] = (index.start, index.stop)
            rec["set_tstamp"], rec["set_poses"], rec["set_disp"], rec["set_intr"] = item[0].clone(), item[2].clone(), item[3], item[4].clone()
            self.poses[index] = item[2]
    v = Video()
    v.counter = N
    v.tstamp = torch.cat([ts, torch.zeros(16)])
    v.poses = torch.cat([poses, torch.zeros(16, 7)])

    class Graph:
        def __init__(self, video, update_op, *a, **k):
            pass

        def add_factors(self, ii, jj):
            rec["calls"].append(("add", torch.as_tensor(ii).clone(), torch.as_tensor(jj).clone()))

        def update(self, t0, t1, motion_only=False):
            rec["calls"].append(("update", t0, t1, motion_only, v.counter))
    monkeypatch.setattr(tf, "FactorGraph", Graph)

    class Net:
        cnet = None
        update = None

        @staticmethod
        def fnet(x):
            return torch.zeros(x.shape[0], x.shape[1], 128, ht // 8, wd // 8)
    filler = tf.PoseTrajectoryFiller(Net, v, device="cpu")
    out = filler._fill(stamps, [torch.zeros(3, ht, wd) for _ in stamps], [torch.tensor([10.0, 10.0, 12.0, 8.0]) for _ in stamps])
    assert np.allclose(rec["set_poses"].numpy(), z["init_poses"], atol=1e-5)
    assert list(rec["set_index"]) == z["set_index"].tolist() and np.allclose(rec["set_tstamp"].numpy(), z["set_tstamp"])
    assert np.allclose(rec["set_intr"].numpy(), z["set_intr"]) and float(rec["set_disp"]) == float(z["set_disp"])
    adds = [c for c in rec["calls"] if c[0] == "add"]
    ups = [c for c in rec["calls"] if c[0] == "update"]
    assert adds[0][1].tolist() == z["add0_ii"].tolist() and adds[0][2].tolist() == z["add0_jj"].tolist()
    assert adds[1][1].tolist() == z["add1_ii"].tolist() and adds[1][2].tolist() == z["add1_jj"].tolist()
    assert [[u[1], u[2], int(u[3]), u[4]] for u in ups] == z["updates"].tolist()
    assert v.counter == int(z["counter_after"]) == N
    assert np.allclose(out[0].data.numpy(), z["returned"], atol=1e-5)


def test_frontend_call_sequence_matches_reference(monkeypatch):
    """DroidFrontend against the reference's own frontend driving the same recording stand-in for FactorGraph
    (tests/golden/gen_golden.py: gen_frontend): the initialisation and four updates, two of which drop a keyframe"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import gen_golden as G
    import pvo_amd.frontend as fe_mod
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_calls.npz"))
    v = G.frontend_video()
    v.counter = 0
    dist = list(G.FRONTEND_DISTANCES)
    G.RecordingGraph.log = []
    v.distance = lambda ii, jj, beta=0.3, bidirectional=True: (G.RecordingGraph.log.append(("dist", list(ii), list(jj), round(float(beta), 6), bidirectional)), torch.tensor([dist.pop(0)]))[1]
    monkeypatch.setattr(fe_mod, "FactorGraph", G.RecordingGraph)
    fe = fe_mod.DroidFrontend(None, v, device="cpu", warmup=5, beta=0.6, frontend_nms=1, keyframe_thresh=2.25,
                              frontend_window=25, frontend_thresh=12.0, frontend_radius=2, max_factors=48)
    snaps = []
    for step in range(10):
        if v.counter < 5 or fe.is_initialized:
            v.counter += 1
        fe()
        snaps.append((v.counter, fe.t1, int(fe.is_initialized), v.poses[:, 0].clone(), v.disps[:, 0, 0].clone(), v.dirty.clone()))
        if not dist and fe.is_initialized and step > 6:
            break
    assert [repr(x) for x in G.RecordingGraph.log] == z["log"].tolist()
    assert [s[0] for s in snaps] == z["counter"].tolist() and [s[1] for s in snaps] == z["t1"].tolist()
    assert [s[2] for s in snaps] == z["init"].tolist()
    assert np.allclose(torch.stack([s[3] for s in snaps]).numpy(), z["poses"]) and np.allclose(torch.stack([s[4] for s in snaps]).numpy(), z["disps"])
    assert np.array_equal(torch.stack([s[5] for s in snaps]).numpy(), z["dirty"])


def test_motion_filter_decisions_match_reference(monkeypatch):
    """MotionFilter.track against the reference's own filter on a mock network (tests/golden/gen_golden.py:
    gen_motion_filter): same frames promoted to keyframes, same pose / depth / intrinsics / features stored for them"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import gen_golden as G
    import pvo_amd.motion_filter as mf_mod
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "motion_filter.npz"))
    ht, wd = 32, 48
    appended = []

    class Video:
        counter = 0

        def append(self, tstamp, pose, disp, intrinsics, fmap, net, inp, segm=None, image=None):
            appended.append((tstamp, pose, disp, intrinsics, fmap, net, inp)); Video.counter += 1

    class FakeCorr:
        def __init__(self, f1, f2, *a, **k):
            pass

        def __call__(self, coords):
            return torch.zeros(1, 1, 196, ht // 8, wd // 8)
    monkeypatch.setattr(mf_mod, "CorrBlock", FakeCorr)
    net = G.MotionNet(ht, wd)
    mf = mf_mod.MotionFilter(net, Video(), thresh=1.75, device="cpu")
    counts = []
    for t, image, intr, segm in G.motion_frames():
        mf.track(t, image, None, intr, segm)
        counts.append(mf.count)
    assert len(appended) == int(z["n_appended"]) and counts == z["counts"].tolist()
    assert [a[0] for a in appended] == z["tstamps"].tolist()
    assert [a[1] is not None for a in appended] == z["has_pose"].tolist() and [a[2] is not None for a in appended] == z["has_disp"].tolist()
    assert torch.equal(appended[0][1].cpu(), torch.tensor([0, 0, 0, 0, 0, 0, 1.0])) and appended[0][2] == 1.0
    assert np.allclose(torch.stack([a[3] for a in appended]).numpy(), z["intr"])
    for k, name in ((4, "fmap_mean"), (5, "net_mean"), (6, "inp_mean")):
        assert np.allclose([float(a[k].float().mean()) for a in appended], z[name], atol=1e-5), name
    assert np.allclose(np.array([[c[2], c[3]] for c in net.calls]), z["op_calls"], atol=1e-5)


def test_depth_video_native_calls_match_reference(monkeypatch):
    """DepthVideo.distance / ba / normalize against the reference's DepthVideo with recording stand-ins for the native
    calls (tests/golden/gen_golden.py: gen_depth_video): argument order, defaults, averaging, clamping"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import gen_golden as G
    import pvo_amd.depth_video as dv
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "depth_video_calls.npz"))
    log = []
    fd, ba = G.make_native_recorder(log)
    monkeypatch.setattr(dv.db, "frame_distance", fd)
    monkeypatch.setattr(dv.db, "ba", ba)
    v = dv.DepthVideo(image_size=(32, 48), buffer=8, device="cpu")

    def set_counter(video, n):
        video.counter = n
    res = G.video_script(v, set_counter, log)
    assert [repr(x) for x in log] == z["log"].tolist()
    for k, t in res.items():
        assert np.allclose(t.numpy(), z[k], atol=1e-6), k


@pytest.mark.gpu
def test_motion_filter_graph_replay_and_fused_encoders_match_the_eager_path(cuda):
    """MotionFilter's per-frame work as captured HIP graphs (pvo_amd/graphs.py) and with the fused encoder layers
    (BasicEncoder.forward_inference) against the eager, module-by-module path on the same frames: the same keyframe decisions, the
    one-step flow magnitudes to fp16-pipeline accuracy, the stored feature maps to 1e-2 of their scale - and the graphs were
    really replayed."""
    from pvo_amd.depth_video import DepthVideo
    from pvo_amd.droid_net import DroidNet
    from pvo_amd.motion_filter import MotionFilter
    ht, wd, n = 128, 160, 14
    frames = list(_textured_stream(n, ht, wd, seed=3))
    torch.manual_seed(0)
    net = DroidNet().to(cuda).eval()
    net.update.half(); net.fnet.half(); net.cnet.half()
    out = {}
    for mode in ("eager", "graphs"):
        video = DepthVideo((ht, wd), buffer=32, device=cuda)
        mf = MotionFilter(net, video, thresh=0.0, device=cuda)
        if mode == "eager":
            mf._features_g.disabled = mf._context_g.disabled = mf._frame_g.disabled = True
            mf.fused_encoders = False
        mags = []
        frame_g = mf._frame_g

        class Spy:
            def __call__(self, *a):
                r = frame_g(*a)
                mags.append(float(r[1]))
                return r
        mf._frame_g = Spy()
        for t, image, intr, segm in frames:
            mf.track(t, image, intrinsics=intr)
        out[mode] = (mags, video.fmaps[:video.counter].float().clone(), video.counter, frame_g.replays)
    assert out["eager"][2] == out["graphs"][2] == n and out["eager"][3] == 0 and out["graphs"][3] >= n - 4
    a, b = torch.tensor(out["eager"][0]), torch.tensor(out["graphs"][0])
    assert torch.allclose(a, b, rtol=3e-2, atol=1e-3), (a, b)
    fa, fb = out["eager"][1], out["graphs"][1]
    assert float((fa - fb).abs().max()) <= 1e-2 * float(fa.abs().max())


@pytest.mark.gpu
def test_graphed_call_skips_only_arguments_the_caller_froze():
    """pvo_amd.graphs.GraphedCall copies every argument into the capture's static buffer on every call, except those the CALLER
    declared frozen - and of those only the same tensor object with an unchanged version counter.  In particular a tensor a
    kernel of this library rewrote through data_ptr() (no version bump) and that is passed again is NOT skipped; if a
    caller freezes such a tensor wrongly, the debug knob finds it."""
    from pvo_amd import config, droid_backends as db
    from pvo_amd.graphs import GraphedCall
    dev = torch.device("cuda:0")
    g = GraphedCall(lambda a, b: a * 2.0 + b, warmup=1, frozen=(1,))
    a, b = torch.arange(8.0, device=dev), torch.ones(8, device=dev)
    with torch.no_grad():
        for _ in range(3):
            out = g(a, b).clone()                      # eager, capture, replay
        assert g.replays >= 1 and torch.equal(out, a * 2 + b)
        n_skip = g.skipped
        assert torch.equal(g(a, b), a * 2 + b) and g.skipped == n_skip + 1      # b frozen, same object, same version: not copied
        b.add_(1.0)                                    # same object, written in place through torch
        assert torch.equal(g(a, b), a * 2 + b)
        for k in range(4):                             # a fresh tensor per call (freed ones may come back at the same address)
            c = torch.full((8,), float(k), device=dev)
            assert torch.equal(g(a, c), a * 2 + c)
            del c
        # argument 0 is not frozen: rewritten behind torch's back (a library kernel writes through data_ptr(): no version bump)
        x = torch.zeros(1, 16, 4, 4, dtype=torch.float16, device=dev)
        y = torch.ones(1, 16, 4, 4, dtype=torch.float16, device=dev)
        h = GraphedCall(lambda u, v: u.float() + v.float(), warmup=1, frozen=(1,))
        for _ in range(3):
            h(x, y)
        ver = x._version
        db.bias_norm_act(torch.full_like(x, 3.0), out=x)           # the library writes x in place
        torch.cuda.synchronize()
        assert x._version == ver and float(x.float().mean()) == 3.0
        assert torch.equal(h(x, y), x.float() + y.float())          # seen: x is copied on every call
        # the same write into a FROZEN argument is the caller breaking its promise; the debug knob reports it
        db.bias_norm_act(torch.full_like(y, 5.0), out=y)
        torch.cuda.synchronize()
        assert not torch.equal(h(x, y), x.float() + y.float())
        config.debug_config("graph_check_skipped", True)
        try:
            with pytest.raises(RuntimeError, match="declared frozen"):
                h(x, y)
        finally:
            config.debug_config("graph_check_skipped", False)


@pytest.mark.gpu
def test_full_sequence_at_240x808_graphs_against_eager_with_segments_and_removals(cuda):
    """BASELINE.json configs[1]'s full-sequence form at the reference driver's input size (tools/test_vo.py's loop: Droid.track per
    frame, Droid.terminate; 240 x 808, panoptic segments, segm_filter on) - 44 frames, every frame a keyframe candidate, a quarter of the
    keyframe updates ending in rm_keyframe by a seeded schedule (DroidFrontend.keyframe_decision): the captured HIP graphs of the
    per-frame work (pvo_amd/graphs.py) against the eager launches of the same kernels.  Same keyframes kept, same removals, the motion
    filter's per-frame flow magnitudes to 16-bit pipeline accuracy, every stored feature map to 1e-2 of its scale; the graphs were
    really replayed.  (Poses are NOT compared tightly: a random-init network turns last-bit differences of its inputs into different
    updates - measured here: 0.03 on poses of magnitude 0.14 after 44 frames between two runs that agree to 1e-3 on every per-frame
    quantity; with trained weights the BA pulls both to the same optimum.  They must be finite and of the same shape.)"""
    import random
    from pvo_amd import config
    from pvo_amd.droid import Droid, default_args
    from pvo_amd.synthetic import drifting_texture_stream
    n = 44
    frames = list(drifting_texture_stream(n, seed=0))
    rng = random.Random(77)
    sched = [rng.random() < 0.25 for _ in range(4 * n)]
    out = {}
    for mode in ("graphs", "eager"):
        config.debug_config("hip_graphs", mode == "graphs")
        try:
            torch.manual_seed(0)
            droid = Droid(default_args(device=str(cuda), image_size=[240, 808], buffer=64, segm_filter=True, thresh=0.8,
                                       filter_thresh=0.0, keyframe_thresh=0.0))
            fe, mf = droid.frontend, droid.filterx
            fe.keyframe_decision = lambda k, dist: sched[k]
            mags, frame_g = [], mf._frame_g

            class Spy:
                replays = property(lambda self: frame_g.replays)

                def __call__(self, *a):
                    r = frame_g(*a)
                    mags.append(float(r[1]))
                    return r
            mf._frame_g = Spy()
            for t, image, intr, segm in frames:
                droid.track(t, image, intrinsics=intr, segments=segm)
            kf = int(droid.video.counter)
            res = dict(kept=droid.video.tstamp[:kf].cpu().clone(), removed=fe.keyframes_removed, poses=droid.video.poses[:kf].cpu().clone(),
                       disps=droid.video.disps[:kf].cpu().clone(), replays=mf._frame_g.replays + mf._context_g.replays, mags=torch.tensor(mags),
                       fmaps=droid.video.fmaps[:kf].float().cpu().clone())
            res["traj"] = torch.from_numpy(droid.terminate(iter(frames), need_inv=True))
            out[mode] = res
            del droid
        finally:
            config.debug_config("hip_graphs", True)
    g, e = out["graphs"], out["eager"]
    assert g["replays"] >= n and e["replays"] == 0
    assert g["removed"] == e["removed"] >= 5 and torch.equal(g["kept"], e["kept"]) and 20 <= g["kept"].shape[0] < n
    assert torch.isfinite(g["traj"]).all() and g["traj"].shape == (n, 7)
    assert torch.allclose(g["mags"], e["mags"], rtol=3e-2, atol=1e-3), (g["mags"], e["mags"])
    assert float((g["fmaps"] - e["fmaps"]).abs().max()) <= 1e-2 * float(e["fmaps"].abs().max())
    assert g["poses"].shape == e["poses"].shape and torch.isfinite(g["poses"]).all() and torch.isfinite(e["traj"]).all()
    print("pose drift between the two runs (random-init network): %.4f on poses of magnitude %.3f"
          % (float((g["poses"] - e["poses"]).abs().max()), float(e["poses"][:, :3].abs().max())))


@pytest.mark.gpu
def test_encoder_head_convolution_is_deterministic_and_matches_fp32(cuda):
    """pvo_conv1x1_planes (the encoders' Conv2d(128, C, 1), extractor.py:139,199) against the fp32 convolution of the same 16-bit operands,
    and the whole fused encoder three times on one frame: identical bits (the vendor library's split-K kernel for this layer was not)."""
    import torch.nn.functional as F
    from pvo_amd import droid_backends as db
    from pvo_amd.modules.extractor import BasicEncoder
    g = torch.Generator().manual_seed(3)
    for cout, hw in ((128, (30, 101)), (256, (30, 101)), (64, (7, 9))):
        x = torch.randn(2, 128, *hw, generator=g).half().to(cuda)
        w = (torch.randn(cout, 128, 1, 1, generator=g) * 0.1).half().to(cuda)
        b = torch.randn(cout, generator=g).half().to(cuda)
        y = db.conv1x1_planes(x, w, b)
        ref = (F.conv2d(x.float(), w.float()).half().float() + b.float()[None, :, None, None]).half()
        assert y.shape == ref.shape and float((y.float() - ref.float()).abs().max()) <= 2e-3 * float(ref.float().abs().max()) + 2e-3
        assert torch.equal(y, db.conv1x1_planes(x, w, b)) and torch.equal(db.conv1x1_planes(x, w, None) , db.conv1x1_planes(x, w))
    for cin, cout, hw in ((32, 64, (120, 404)), (64, 128, (60, 202)), (32, 64, (7, 9))):      # the residual blocks' strided shortcuts
        x = torch.randn(1, cin, *hw, generator=g).half().to(cuda)
        w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).half().to(cuda)
        b = torch.randn(cout, generator=g).half().to(cuda)
        y = db.conv1x1_planes(x, w, b, stride=2)
        ref = (F.conv2d(x.float(), w.float(), stride=2).half().float() + b.float()[None, :, None, None]).half()
        assert y.shape == ref.shape and float((y.float() - ref.float()).abs().max()) <= 2e-3 * float(ref.float().abs().max()) + 2e-3
    with pytest.raises(db.PvoHipError):
        db.conv1x1_planes(torch.zeros(1, 48, 4, 4, dtype=torch.float16, device=cuda), torch.zeros(64, 48, dtype=torch.float16, device=cuda))
    # the frame's normalisation as one kernel: bit-identical to the element-wise sequence of motion_filter.py:52-54 + the encoder's cast
    mean, std = torch.tensor([0.485, 0.456, 0.406], device=cuda)[:, None, None], torch.tensor([0.229, 0.224, 0.225], device=cuda)[:, None, None]
    for dt in (torch.int32, torch.uint8, torch.float32):
        img = torch.randint(0, 256, (3, 37, 53), generator=g).to(dt).to(cuda)
        ref = (((img.flip(0)[None].float() / 255.0) - mean) / std).half()
        assert torch.equal(db.frame_normalise(img, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)), ref)
    torch.manual_seed(0)
    for norm, dim in (("instance", 128), ("none", 256)):
        enc = BasicEncoder(output_dim=dim, norm_fn=norm).to(cuda).half().eval()
        img = torch.randn(1, 1, 3, 240, 808, generator=g).to(cuda)
        with torch.no_grad():
            outs = [enc.forward_inference(img) for _ in range(3)]
            ref = enc(img.half())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert float((outs[0].float() - ref.float()).abs().max()) <= 2e-2 * float(ref.float().abs().max())


@pytest.mark.gpu
def test_pipelined_tracker_gives_the_same_video_and_trajectory_bit_for_bit(cuda):
    """Droid(args.pipelined=True) - frame t + 1's encoder graph launched before the second half of keyframe t's frontend update, the frame
    uploaded on the second stream - against the reference's order, 240 x 808 with segments, a quarter of the keyframe updates ending in
    rm_keyframe (seeded), a mix of filtered and kept frames: the same operations on the same data in the same dependency order, so
    keyframes, poses, depths and the filled-in trajectory are IDENTICAL, not close.  Between calls the pipelined video lags (an update is
    pending after a keyframe), get_traj() completes it.  (Comparing bits needs a run that repeats itself: a third pass in the reference's
    order pins that too - it did not before the encoders' last layer left the vendor library's split-K kernel, pvo_conv1x1_planes.)"""
    import random
    from pvo_amd.droid import Droid, default_args
    from pvo_amd.synthetic import drifting_texture_stream
    n = 40
    frames = list(drifting_texture_stream(n, seed=0))
    rng = random.Random(5)
    sched = [rng.random() < 0.25 for _ in range(4 * n)]
    out = {}
    for mode in (False, True, "again", "no prefetch"):
        torch.manual_seed(0)
        droid = Droid(default_args(device=str(cuda), image_size=[240, 808], buffer=64, segm_filter=True, thresh=0.8,
                                   filter_thresh=0.2026, keyframe_thresh=0.0, pipelined=mode is True))
        fe = droid.frontend
        fe.keyframe_decision = lambda k, dist: sched[k]
        fe.prefetch = mode != "no prefetch"          # (the proximity distances launched ahead of the context encoder, or where the reference computes them)
        pending, used, pf = 0, 0, fe.graph.prefetch_proximity

        def spy(*a, **kw):
            nonlocal used
            used += 1
            return pf(*a, **kw)
        fe.graph.prefetch_proximity = spy
        for t, image, intr, segm in frames:
            droid.track(t, image, intrinsics=intr, segments=segm)
            pending += int(fe.update_pending)
        mid = torch.from_numpy(droid.get_traj()).clone()                     # (flushes)
        assert not fe.update_pending
        kf = int(droid.video.counter)
        res = dict(kept=droid.video.tstamp[:kf].cpu().clone(), removed=fe.keyframes_removed, poses=droid.video.poses[:kf + 1].cpu().clone(),
                   disps=droid.video.disps[:kf + 1].cpu().clone(), pending=pending, mid=mid, updates=fe.count, prefetched=used)
        res["traj"] = torch.from_numpy(droid.terminate(iter(frames), need_inv=True))
        out[mode] = res
        del droid
    a, b = out[False], out["again"]
    for k in ("kept", "poses", "disps", "mid", "traj"):                      # the sequence itself is reproducible bit for bit from run to run
        assert torch.equal(a[k], b[k]), ("two runs in the reference's order differ", k)
    b = out["no prefetch"]
    assert b["prefetched"] == 0 and a["prefetched"] >= a["updates"] - 2      # every keyframe update but the first read prefetched distances
    for k in ("kept", "poses", "disps", "mid", "traj"):
        assert torch.equal(a[k], b[k]), ("prefetched proximity distances change the result", k)
    a, b = out[False], out[True]
    assert a["pending"] == 0 and b["pending"] >= 10                          # the pipelined tracker really left updates in flight
    assert a["updates"] == b["updates"] >= 15 and a["removed"] == b["removed"] >= 3
    assert torch.equal(a["kept"], b["kept"]) and a["kept"].shape[0] < n - 3  # the filter dropped frames, the same ones
    for k in ("poses", "disps", "mid", "traj"):
        assert torch.equal(a[k], b[k]), k
    assert torch.isfinite(a["traj"]).all() and a["traj"].shape == (n, 7)


def test_graphed_call_is_a_plain_call_off_the_gpu():
    """CPU tensors, gradients enabled or no arguments: pvo_amd.graphs.GraphedCall just calls through (nothing is captured)"""
    from pvo_amd.graphs import GraphedCall
    calls = []
    g = GraphedCall(lambda a, b: (calls.append(1), a + b)[1], warmup=0)
    a, b = torch.ones(3), torch.arange(3.0)
    with torch.no_grad():
        for _ in range(4):
            assert torch.equal(g(a, b), a + b)
    assert len(calls) == 4 and g.replays == 0 and not g.cache


def test_host_knobs_are_set_by_code_not_by_the_environment(monkeypatch):
    """pvo_amd.config: the host layer's test / A-B knobs (two ranks with different environments must not run different
    schedules).  An environment variable of the old name changes nothing; an unknown knob raises; "se3_torch" reaches geom.se3."""
    import importlib
    from pvo_amd import config
    from pvo_amd.geom import se3
    monkeypatch.setenv("PVO_HIP_GRAPHS", "0"); monkeypatch.setenv("PVO_SE3_TORCH", "1"); monkeypatch.setenv("PVO_CONV128_WIDE", "0")
    importlib.reload(config)
    assert config.get("hip_graphs") is True and config.get("conv128_wide") is True and se3.FORCE_TORCH is False
    with pytest.raises(KeyError):
        config.debug_config("no_such_knob", 1)
    config.debug_config("se3_torch", True)
    try:
        assert se3.FORCE_TORCH is True
    finally:
        config.debug_config("se3_torch", False)
    assert se3.FORCE_TORCH is False
    for mod in ("pvo_amd/graphs.py", "pvo_amd/modules/update.py", "pvo_amd/geom/se3.py", "pvo_amd/config.py"):
        assert "os.environ" not in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mod)).read(), mod


def test_kept_feature_maps_are_only_reused_for_the_frame_they_came_from():
    """DepthVideo.remember_features / recall_features: the trajectory filler reuses a tracked frame's feature map only if the
    image it is handed under that time stamp is the image the map was computed from; an entry is released when it is asked for"""
    from pvo_amd.depth_video import DepthVideo
    v = DepthVideo((64, 96), buffer=4, device="cpu")
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (3, 64, 96), generator=g, dtype=torch.int32)
    other = img.clone(); other[:, 20:40, 30:60] += 7
    fmap = torch.randn(1, 128, 8, 12, generator=g).half()
    v.remember_features(3.0, fmap, img)
    assert v.recall_features(3.0, other) is None and not v.frame_fmaps            # another image under the same time stamp: re-encode
    v.remember_features(3.0, fmap, img)
    got = v.recall_features(3.0, img)
    assert got is not None and torch.equal(got, fmap) and v.recall_features(3.0, img) is None and v._frame_fmaps_bytes == 0
    v.remember_features(4.0, fmap, img); v.remember_features(5.0, fmap)
    assert v.recall_features(5.0, img) is None                                     # kept without a fingerprint: never trusted
    v.forget_features()
    assert not v.frame_fmaps and v._frame_fmaps_bytes == 0


def test_frontend_keyframe_decision_hook_overrides_the_distance_test():
    """DroidFrontend.keyframe_decision (measurement / test hook): the removal branch of droid_frontend.py:54-58 driven by a schedule"""
    from pvo_amd.frontend import DroidFrontend

    class G:
        corr = None
        _ii_h = [0]
        calls = []
        def add_proximity_factors(self, *a, **k): pass
        def update(self, *a, **k): self.calls.append("update")
        def rm_keyframe(self, ix): self.calls.append(("rm", ix))

    class V:
        counter = 10
        poses = torch.zeros(16, 7); disps = torch.ones(16, 2, 2); dirty = torch.zeros(16, dtype=torch.bool)
        def distance(self, ii, jj, **k): return torch.tensor([5.0])
    fe = DroidFrontend.__new__(DroidFrontend)
    fe.video, fe.graph = V(), G()
    fe.t0, fe.t1, fe.count, fe.is_initialized = 0, 8, 0, True
    fe.max_age, fe.iters1, fe.iters2, fe.beta, fe.frontend_nms = 25, 4, 2, 0.3, 1
    fe.keyframe_thresh, fe.frontend_window, fe.frontend_thresh, fe.frontend_radius = 1.0, 25, 16.0, 2
    fe.keyframes_removed = 0
    seen = []
    fe.keyframe_decision = lambda k, d: (seen.append((k, d)), True)[1]
    fe._update()
    assert seen == [(1, 5.0)] and ("rm", 7) in fe.graph.calls and fe.video.counter == 9 and fe.t1 == 8 and fe.keyframes_removed == 1
    fe.keyframe_decision = None                                                    # the reference's test: 5.0 >= 1.0 -> kept, two more updates
    n = len(fe.graph.calls)
    fe._update()
    assert fe.graph.calls[n:].count("update") == 6 and fe.video.counter == 9 and fe.t1 == 9 and fe.keyframes_removed == 1
