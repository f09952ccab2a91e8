"""FactorGraph.update glue against the reference's FactorGraph.update, both run with the same
recorded stand-ins for reproject / update operator / BA (fixtures from tests/golden/gen_golden.py).
The first test runs the PyTorch glue on the CPU with the three native calls stubbed; the `gpu` tests run the HIP glue
kernels of the native update against the same fixtures on the device."""
import os

import numpy as np
import pytest
import torch

from pvo_amd.factor_graph import FactorGraph

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,segm_filter", [("plain", False), ("segm", True)])
def test_update_glue_matches_reference(name, segm_filter):
    d = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "factor_graph_glue_%s.npz" % name)).items()}
    E, ht, wd = d["coords1"].shape[1:4]
    cap = {}

    class Video:
        pass
    v = Video()
    v.ht, v.wd, v.disps = ht * 8, wd * 8, torch.ones(4, ht, wd)
    v.segm_filter, v.thresh, v.max_segments = segm_filter, 0.5, 16
    v.reproject = lambda a, b: (d["coords1"].clone(), torch.ones(1, E, ht, wd, 1))

    def ba(target, weight, eta, ii, jj, t0, t1, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        cap.update(target=target, weight=weight, eta=eta, ii=ii, jj=jj, t0=t0, t1=t1, itrs=itrs, lm=lm, ep=ep)
    v.ba = ba

    def update_op(net, inp, corr, motn, ii, jj, flag):
        cap["motn"] = motn
        return torch.zeros(1, E, 128, ht, wd), d["delta"], d["weight_out"], d["damping_out"], {}, d["delta_m"]

    fg = FactorGraph(v, update_op, device="cpu")
    fg.ii, fg.jj = d["ii"].clone(), d["jj"].clone()
    fg._ii_h, fg._jj_h, fg._age_h = d["ii"].tolist(), d["jj"].tolist(), [0] * E
    fg.age = torch.zeros(E, dtype=torch.long)
    fg.net = fg.inp = torch.zeros(1, E, 128, ht, wd)
    fg.segm = d["segm"].clone()
    fg.target_cam, fg.weight = d["target_cam"].clone(), d["weight0"].clone()
    fg.raw_mask, fg.delta_dy = d["raw_mask"].clone(), d["delta_dy"].clone()
    fg.corr = lambda c: torch.zeros(1, E, 196, ht, wd)
    fg.update(None, 4, itrs=2)

    eq = lambda a, b: torch.allclose(a.float(), b.float(), atol=1e-6, rtol=1e-6)
    assert eq(cap["motn"], d["motn"])
    assert eq(cap["target"], d["ba_target"]) and eq(cap["weight"], d["ba_weight"]) and eq(cap["eta"], d["ba_eta"])
    assert torch.equal(cap["ii"], d["ba_ii"]) and torch.equal(cap["jj"], d["ba_jj"])
    assert cap["t0"] == int(d["ba_t0"]) and cap["itrs"] == int(d["ba_itrs"]) and cap["lm"] == 1e-4 and cap["ep"] == 0.1
    assert eq(fg.target_cam, d["out_target_cam"]) and eq(fg.weight, d["out_weight"])
    assert eq(fg.raw_mask, d["out_raw_mask"]) and eq(fg.delta_dy, d["out_delta_dy"])
    assert eq(fg.full_flow, d["out_full_flow"]) and eq(fg.damping, d["out_damping"])
    assert torch.equal(fg.age, d["out_age"])
    if segm_filter:      # the vote must actually have changed something in this fixture
        plain = np.load(os.path.join(G, "factor_graph_glue_plain.npz"))
        assert not np.allclose(plain["out_weight"], d["out_weight"].numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("name,segm_filter", [("plain", False), ("segm", True)])
def test_hip_glue_kernels_match_reference_fixture(cuda, name, segm_filter):
    """The kernels the native graph update runs around the operator - pvo_graph_motion, pvo_segment_hist, pvo_graph_post,
    pvo_eta_head's damping bookkeeping - on the DEVICE against what the reference's own FactorGraph.update produced
    (factor_graph.py:231-300; fixtures factor_graph_glue_{plain,segm}.npz).  The operator hands its outputs over in
    16 bits, so the fixture's fp32 head outputs are rounded to fp16 on the way in: tolerances are that rounding, and
    pixels whose updated mask logit is within it of the 0.5 threshold are excluded from the thresholded quantities."""
    from pvo_amd import droid_backends as db
    d = {k: torch.from_numpy(v).to(cuda) for k, v in np.load(os.path.join(G, "factor_graph_glue_%s.npz" % name)).items()}
    E, ht, wd = d["coords1"].shape[1:4]
    coords1 = d["coords1"].contiguous()
    # motion features (:233-237)
    motn = db.graph_motion(d["target_cam"].contiguous(), coords1, d["delta_dy"].contiguous(), d["raw_mask"].contiguous(), torch.float16)
    assert motn.shape == d["motn"].shape
    assert torch.allclose(motn.float(), d["motn"], atol=1e-6, rtol=1e-3)
    # operator outputs as the HIP heads kernel hands them over: [E,8,H,W] channels-last, delta | delta_dy | weight | delta_mask
    heads = torch.cat([d["delta"][0], d["weight_out"][0], d["delta_m"][0]], -1).half().contiguous().permute(0, 3, 1, 2)
    raw = d["raw_mask"].clone().contiguous()
    vote = None
    if segm_filter:
        S = 16
        segm = d["segm"][0, :, 0].int().contiguous()
        assert int(segm.max()) < S
        tot, dyn = db.segment_hist(segm, raw, heads, S, 0.5)
        vote = (segm, tot, dyn, 0.5)
    tb, wb = torch.empty(E, 2, ht, wd, device=cuda), torch.empty(E, 2, ht, wd, device=cuda)
    target, ddy, weight, flow = db.graph_post(coords1, heads, raw, tb, wb, 0.5, vote=vote)
    h16 = lambda t: 2.0 ** -10 * t.abs().max().item() + 1e-6            # one fp16 rounding of the largest input
    assert torch.allclose(raw, d["out_raw_mask"], atol=h16(d["delta_m"]))
    assert torch.allclose(target, d["out_target_cam"], atol=h16(d["delta"]))
    safe = ((d["out_raw_mask"].abs() > 2 * h16(d["delta_m"])).all(-1, keepdim=True)).float()      # away from the threshold
    assert safe.mean() > 0.95
    for got, want, scale in ((ddy, d["out_delta_dy"], d["delta"]), (weight, d["out_weight"], d["weight_out"]),
                             (flow, d["out_full_flow"], d["delta"])):
        assert (((got - want).abs() * safe).max() < h16(scale)), (name, ((got - want).abs() * safe).max())
    # the BA's [E,2,H,W] layouts (:292-300)
    assert torch.allclose(tb, d["ba_target"], atol=h16(d["delta"]))
    assert (((wb - d["ba_weight"]).abs() * safe[0].permute(0, 3, 1, 2)).max() < h16(d["weight_out"]))
    if segm_filter:      # the vote forced whole segments: compare with the vote switched off
        _, _, w_off, _ = db.graph_post(coords1, heads, d["raw_mask"].clone().contiguous(), tb, wb, 0.5)
        assert (w_off - weight).abs().max() > 0.1
    # damping bookkeeping (:281-297): an eta head whose convolution reproduces the fixture's operator output
    K = d["damping_out"].shape[1]
    v = torch.log(torch.expm1((100.0 * d["damping_out"][0]).double())).float()            # softplus^-1(eta / 0.01)
    x = torch.zeros(K, ht, wd, 128, device=cuda)
    x[..., 0] = v
    wt = torch.zeros(9, 128, device=cuda); wt[4, 0] = 1.0                                  # centre tap, channel 0
    src = sorted(set(d["ii"].tolist()))
    rows = sorted(set(d["ba_ii"].tolist()))
    where = {f: k for k, f in enumerate(src)}
    damping = 1e-6 * torch.ones(d["out_damping"].shape, device=cuda)
    eta = db.eta_head(x.half().permute(0, 3, 1, 2), wt.half(), torch.zeros(1, device=cuda), torch.tensor(rows, device=cuda),
                      torch.tensor([where.get(f, -1) for f in rows], dtype=torch.int32, device=cuda), damping, 1e-7)
    assert torch.allclose(damping, d["out_damping"], rtol=4e-3, atol=1e-7)
    assert torch.allclose(eta, d["ba_eta"], rtol=4e-3, atol=1e-7)


@pytest.mark.gpu
def test_native_update_survives_edge_changes(cuda):
    """pvo_graph_update across an edge change (drop the newest keyframe's edges with storage, add some back): the
    per-edge-set caches (index tensors, BA plan, inactive rows, static GRU terms) follow the edge set, and the native path
    agrees with the PyTorch-glue path step by step"""
    import bench
    res = []
    for fused in (True, False):
        video, graph = bench.make_window(cuda, seed=3)
        graph.fused_glue = fused
        for _ in range(2):
            graph.update(None, None, use_inactive=True)
        newest = bench.NKF - 1
        m = [(i == newest or j == newest) for i, j in zip(graph._ii_h, graph._jj_h)]
        pairs = [(i, j) for i, j in zip(graph._ii_h, graph._jj_h) if i == newest or j == newest]
        graph.rm_factors(m, store=True)
        graph.add_factors([p[1] for p in pairs[:4]], [p[0] for p in pairs[:4]])
        graph.update(None, None, use_inactive=True)
        torch.cuda.synchronize()
        res.append(dict(poses=video.poses.clone(), disps=video.disps.clone(), net=graph.net.float().clone(),
                        target=graph.target_cam.clone(), raw=graph.raw_mask.clone(), damping=graph.damping.clone(),
                        age=graph.age.clone(), age_h=list(graph._age_h)))
    a, b = res
    assert a["age_h"] == b["age_h"] and torch.equal(a["age"], b["age"]) and a["age_h"] == a["age"].tolist()
    for k in ("poses", "disps", "net", "target", "raw", "damping"):
        dd = (a[k].float() - b[k].float()).abs()
        print(k, "max %.3g mean %.3g" % (dd.max().item(), dd.mean().item()))
        assert dd.mean().item() < 2e-3, k


@pytest.mark.gpu
def test_native_global_update_on_resident_volumes(cuda):
    """FactorGraph.update_lowmem with corr_impl="volume" (the whole global graph in one pvo_graph_update per step, volumes
    resident in HBM) against the reference's formulation of the same update (alt-corr lookup, operator in 8-frame chunks,
    PyTorch glue, DepthVideo.ba) on the same window; and its edge-sharded variant (operator native with itrs=0, BA through
    ShardedBA) against the unsharded call - integer accumulation makes those two bit-identical."""
    import bench
    from pvo_amd.parallel import ShardedBA
    nkf = 12
    ii = [i for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 2]
    jj = [j for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 2]
    res = {}
    for name, corr_impl, fused, sharded in (("native", "volume", True, None), ("sharded", "volume", True, ShardedBA()),
                                            ("alt", "alt", False, None)):
        video, graph = bench.make_window(cuda, seed=5, NKF=nkf, buffer=16, corr_impl=corr_impl, add_edges=False, max_factors=-1)
        video.counter = nkf
        graph.fused_glue = fused
        graph.add_factors(list(ii), list(jj))
        assert (corr_impl == "volume") == (graph.P_zr is not None and graph._fused_ok())
        p0 = video.poses.clone()
        graph.update_lowmem(steps=2, sharded=sharded)
        torch.cuda.synchronize()
        assert (video.poses - p0).abs().max() > 1e-4
        res[name] = dict(poses=video.poses.clone(), disps=video.disps.clone(), net=graph.net.float().clone(),
                         target=graph.target_cam.clone(), raw=graph.raw_mask.clone(), damping=graph.damping.clone())
    for k in res["native"]:
        assert torch.equal(res["native"][k], res["sharded"][k]), k
        dd = (res["native"][k] - res["alt"][k]).abs()
        print(k, "max %.3g mean %.3g" % (dd.max().item(), dd.mean().item()))
        assert dd.mean().item() < 2e-3, k


def test_update_lowmem_glue_matches_reference(monkeypatch):
    """FactorGraph.update_lowmem against the reference's own update_lowmem run with the same recorded stand-ins
    (tests/golden/gen_golden.py: gen_lowmem_glue): chunking by source frame, motion features, damping, BA arguments."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from gen_golden import LOWMEM_EDGES, lowmem_records
    import pvo_amd.modules.corr as corr_mod
    from pvo_amd.factor_graph import FactorGraph
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lowmem_glue.npz"))
    rec = lowmem_records()
    E, ht, wd, F = rec["E"], rec["ht"], rec["wd"], rec["F"]
    ii_l, jj_l = LOWMEM_EDGES
    cap = {"motn": [], "ba": [], "calls": []}
    step = {"k": -1}

    class Video:
        pass
    v = Video()
    v.ht, v.wd, v.counter = ht * 8, wd * 8, F
    v.disps, v.dirty = torch.ones(F, ht, wd), torch.zeros(F, dtype=torch.bool)
    v.segm_filter, v.thresh = False, 0.5
    v.fmaps, v.inps, v.nets = rec["fmaps"].permute(0, 2, 3, 1).contiguous(), rec["inps"], torch.zeros(F, 128, ht, wd)
    v.segms = torch.zeros(F, 1, ht, wd, dtype=torch.int)

    def reproject(a, b):
        step["k"] += 1
        return rec["coords1"][step["k"]].clone(), torch.ones(1, E, ht, wd, 1)
    v.reproject = reproject
    v.ba = lambda target, weight, eta, ii_, jj_, t0, t1, itrs=2, lm=1e-4, ep=0.1, motion_only=False: cap["ba"].append(
        dict(target=target.clone(), weight=weight.clone(), eta=eta.clone(), args=[t0, t1, itrs, lm, ep, float(motion_only)]))

    class FakeAlt:
        def __init__(self, fmaps, *a, **k):
            pass

        def __call__(self, coords, ii_, jj_):
            return torch.zeros(1, ii_.shape[0], 196, ht, wd)
    monkeypatch.setattr(corr_mod, "AltCorrBlock", FakeAlt)

    def update_op(net, inp, corr, motn, ii_, jj_, flag):
        k = step["k"]
        sel = torch.tensor([e for e in range(E) if (ii_l[e] // 8) == (int(ii_[0]) // 8)])
        cap["motn"].append(motn.clone()); cap["calls"].append(sel.clone())
        assert torch.equal(inp[0], rec["inps"][ii_])                   # context features of the source frames
        frames = torch.unique(ii_)
        return (rec["net_out"][k][:, sel], rec["delta"][k][:, sel], rec["weight_out"][k][:, sel],
                rec["damp_table"][k][frames][None], {}, rec["delta_m"][k][:, sel])
    fg = FactorGraph(v, update_op, device="cpu", corr_impl="alt")
    fg.ii, fg.jj = torch.tensor(ii_l), torch.tensor(jj_l)
    fg._ii_h, fg._jj_h, fg._age_h = list(ii_l), list(jj_l), [0] * E
    fg.age = torch.zeros(E, dtype=torch.long)
    fg.net = rec["net"].clone()
    fg.target_cam, fg.weight = rec["target_cam"].clone(), rec["weight0"].clone()
    fg.raw_mask, fg.delta_dy = rec["raw_mask"].clone(), rec["delta_dy"].clone()
    fg.update_lowmem(steps=2)

    assert len(cap["calls"]) == int(z["n_calls"]) and len(cap["ba"]) == 2
    for n in range(len(cap["calls"])):
        assert np.array_equal(cap["calls"][n].numpy(), z["call_edges_%d" % n])
        assert np.allclose(cap["motn"][n].numpy(), z["motn_%d" % n], atol=1e-6)
    for n, b in enumerate(cap["ba"]):
        assert np.allclose(b["target"].numpy(), z["ba_target_%d" % n], atol=1e-6)
        assert np.allclose(b["weight"].numpy(), z["ba_weight_%d" % n], atol=1e-6)
        assert np.allclose(b["eta"].numpy(), z["ba_eta_%d" % n], atol=1e-9)
        assert np.allclose(np.array(b["args"]), z["ba_args_%d" % n])
    for name, t in (("out_target_cam", fg.target_cam), ("out_weight", fg.weight), ("out_raw_mask", fg.raw_mask),
                    ("out_delta_dy", fg.delta_dy), ("out_damping", fg.damping)):
        assert np.allclose(t.numpy(), z[name], atol=1e-6), name
    assert np.allclose(fg.net.numpy(), z["out_net"].astype(np.float32), atol=2e-3)
    assert np.array_equal(v.dirty.numpy(), z["dirty"])


def test_proximity_factor_selection_matches_reference():
    """add_proximity_factors against the reference's own greedy selection on a recorded distance matrix
    (tests/golden/gen_golden.py: gen_proximity): same edges, same order, same `remove` flag"""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from gen_golden import proximity_case
    from pvo_amd.factor_graph import FactorGraph
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "proximity_factors.npz"))
    t, d, have, bad, inac = proximity_case()
    assert np.array_equal(d.numpy(), z["d"])

    class Video:
        pass
    v = Video()
    v.ht, v.wd, v.counter = 64, 64, t
    v.disps = torch.ones(t, 8, 8)
    v.segm_filter = False
    v.distance = lambda ii, jj, beta=0.3: d[torch.as_tensor(ii).long(), torch.as_tensor(jj).long()].clone()
    for n in range(3):
        t0, t1, rad, nms, thresh = [float(x) for x in z["args_%d" % n]]
        fg = FactorGraph(v, None, device="cpu")
        fg.native_select = bool(n != 1)                  # (case 1 through the numpy form, 0 and 2 through pvo_proximity_select)
        fg._ii_h, fg._jj_h = list(have[0]), list(have[1])
        fg.ii, fg.jj = torch.tensor(have[0]), torch.tensor(have[1])
        fg.ii_bad, fg.jj_bad = torch.tensor(bad[0]), torch.tensor(bad[1])
        fg._ii_inac_h, fg._jj_inac_h = list(inac[0]), list(inac[1])
        got = {}
        fg.add_factors = lambda ii, jj, remove=False: got.update(ii=list(ii), jj=list(jj), remove=remove)
        fg.add_proximity_factors(int(t0), int(t1), rad=int(rad), nms=int(nms), beta=0.3, thresh=thresh, remove=bool(n))
        assert got["ii"] == z["ii_%d" % n].tolist() and got["jj"] == z["jj_%d" % n].tolist(), n
        assert int(got["remove"]) == int(z["remove_%d" % n])


def _greedy_selection_loops(t, t0, t1, rad, nms, thresh, d, have):
    """the reference's selection (factor_graph.py:372-429) loop for loop on a host list - the form round 4 shipped and the
    fixture above pins - as the yardstick for the array formulation"""
    ix, jx = list(range(t0, t)), list(range(t1, t))
    ii = [i for i in ix for _ in jx]
    jj = [j for _ in ix for j in jx]
    d = list(d)
    inf, nj = float("inf"), t - t1
    for k, (i, j) in enumerate(zip(ii, jj)):
        if i - rad < j or d[k] > 100:
            d[k] = inf

    def suppress(i, j):
        for di in range(-nms, nms + 1):
            for dj in range(-nms, nms + 1):
                if abs(di) + abs(dj) <= max(min(abs(i - j) - 2, nms), 0):
                    i1, j1 = i + di, j + dj
                    if t0 <= i1 < t and t1 <= j1 < t:
                        d[(i1 - t0) * nj + (j1 - t1)] = inf
    for i, j in have:
        if abs(i - j) > 2:
            suppress(i, j)
    es = []
    for i in range(t0, t):
        for j in range(i + 1, min(i + rad + 1, t)):
            es += [(i, j), (j, i)]
    for k in sorted(range(len(d)), key=lambda k: d[k]):
        if d[k] > thresh:
            continue
        i, j = ii[k], jj[k]
        es += [(i, j), (j, i)]
        suppress(i, j)
    return es


@pytest.mark.parametrize("native", (True, False))
@pytest.mark.parametrize("seed", range(6))
def test_proximity_factor_selection_array_form_equals_the_loops(seed, native):
    """random distance matrices (ties, values above 100, windows of the frontend's and the backend's shapes) and random existing
    edges: the library's host function (pvo_proximity_select, the default) and the numpy array formulation both select the same edges in
    the same order as the reference's loops"""
    import numpy as np
    from pvo_amd.factor_graph import FactorGraph
    g = np.random.default_rng(seed)
    t = int(g.integers(8, 40))
    t0, t1 = (int(g.integers(0, t - 2)), int(g.integers(0, t - 2))) if seed % 2 else (0, 0)
    rad, nms = int(g.integers(1, 4)), int(g.integers(0, 4))
    thresh = float(g.uniform(2.0, 30.0))
    full = np.round(g.uniform(0.0, 60.0, (t, t)), 1).astype(np.float32)        # one decimal: ties
    full[g.uniform(size=(t, t)) < 0.05] = 250.0
    n_have = int(g.integers(0, 60))
    have = [(int(a), int(b)) for a, b in zip(g.integers(0, t, n_have), g.integers(0, t, n_have))]
    third = len(have) // 3

    class Video:
        pass
    v = Video()
    v.ht, v.wd, v.counter, v.segm_filter = 64, 64, t, False
    v.disps = torch.ones(t, 8, 8)
    dmat = torch.from_numpy(full)
    v.distance = lambda ii, jj, beta=0.3: dmat[torch.as_tensor(ii).long(), torch.as_tensor(jj).long()].clone()
    fg = FactorGraph(v, None, device="cpu")
    assert fg.native_select
    fg.native_select = native
    a, b, c = have[:third], have[third:2 * third], have[2 * third:]
    fg._ii_h, fg._jj_h = [e[0] for e in a], [e[1] for e in a]
    fg.ii_bad, fg.jj_bad = torch.tensor([e[0] for e in b], dtype=torch.long), torch.tensor([e[1] for e in b], dtype=torch.long)
    fg._ii_inac_h, fg._jj_inac_h = [e[0] for e in c], [e[1] for e in c]
    got = {}
    fg.add_factors = lambda ii, jj, remove=False: got.update(es=list(zip(ii, jj)))
    fg.add_proximity_factors(t0, t1, rad=rad, nms=nms, beta=0.3, thresh=thresh)
    ix, jx = list(range(t0, t)), list(range(t1, t))
    d = [float(full[i, j]) for i in ix for j in jx]
    want = _greedy_selection_loops(t, t0, t1, rad, nms, thresh, d, a + b + c)
    assert got.get("es", []) == want


def test_edge_bookkeeping_matches_reference():
    """add_neighborhood_factors / add_factors (duplicates, age-based eviction with storage) / rm_keyframe / rm_factors
    against the reference's FactorGraph on a mock video (tests/golden/gen_golden.py: gen_bookkeeping), state compared after
    every step - including the reference's position-indexed eviction mask and rm_keyframe's buffer moves"""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from gen_golden import bookkeeping_script, bookkeeping_video
    from pvo_amd.factor_graph import FactorGraph
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "edge_bookkeeping.npz"))
    v = bookkeeping_video()
    v.fmaps = v.fmaps.permute(0, 2, 3, 1).contiguous()               # the build keeps feature maps channels-last
    fg = FactorGraph(v, None, device="cpu", corr_impl="alt", max_factors=10)

    def set_age(f, ages):
        f.age = torch.tensor(ages)
        f._age_h = list(ages)
    snaps = bookkeeping_script(fg, set_age)
    assert len(snaps) == int(z["n_snaps"])
    for n, sn in enumerate(snaps):
        for k, t in sn.items():
            assert np.array_equal(t.numpy(), z["%s_%d" % (k, n)]), (k, n, t, z["%s_%d" % (k, n)])
        # the host mirrors track the device lists
        assert fg._ii_h == fg.ii.tolist() and fg._jj_h == fg.jj.tolist() and fg._age_h == fg.age.tolist() or n < len(snaps) - 1
    assert fg._ii_h == fg.ii.tolist() and fg._jj_h == fg.jj.tolist() and fg._age_h == fg.age.tolist()
    assert fg._ii_inac_h == fg.ii_inac.tolist() and fg._jj_inac_h == fg.jj_inac.tolist()
    assert np.array_equal(v.poses.numpy(), z["poses"]) and np.array_equal(v.disps.numpy(), z["disps"])
    assert np.array_equal(v.nets[:, 0, 0, 0].numpy(), z["nets"]) and np.array_equal(v.fmaps[:, 0, 0, 0].numpy(), z["fmaps"])
    assert np.array_equal(v.segms[:, 0, 0, 0].numpy(), z["segms"])


def test_edge_rows_follow_plain_indexing_and_concatenation():
    """_EdgeRows (the fixed-capacity row buffers behind net / target_cam / weight / raw_mask / delta_dy / segm) against the
    reference's formulation of the same bookkeeping - t[:, keep] on a drop, torch.cat on an add - over a random sequence
    of drops (suffix, prefix, scattered, everything), adds (within and beyond the capacity) and foreign assignments."""
    import random
    from pvo_amd.factor_graph import _EdgeRows
    rng = random.Random(3)
    g = torch.Generator().manual_seed(5)
    rows = _EdgeRows(8)
    model = torch.zeros(0, 3, 2)
    live = model.clone()
    moved = 0
    for step in range(200):
        op = rng.choice(["suffix", "prefix", "scatter", "all", "add", "add", "add_big", "foreign", "stale"])
        E = model.shape[0]
        if op in ("suffix", "prefix", "scatter", "all") and E:
            k = rng.randint(1, E)
            if op == "suffix":
                keep_l = list(range(E - k))
            elif op == "prefix":
                keep_l = list(range(k, E))
            elif op == "scatter":
                keep_l = sorted(rng.sample(range(E), E - k))
            else:
                keep_l = []
            before = live.data_ptr()
            live = rows.keep(live, keep_l, lambda: torch.tensor(keep_l, dtype=torch.long))
            model = model[keep_l] if keep_l else model[:0]
            if op == "suffix" and rows.owns(live) and live.data_ptr() == before:
                moved += 1                                            # a dropped suffix moves nothing
        elif op in ("add", "add_big"):
            n = rng.randint(1, 4) if op == "add" else rng.randint(10, 30)
            new = torch.randn(n, 3, 2, generator=g)
            live = rows.append(live, n, lambda dst: dst.copy_(new))
            model = torch.cat([model, new])
        elif op == "foreign" and E:                                   # a caller assigns its own tensor (the PyTorch update path)
            model = torch.randn(E, 3, 2, generator=g)
            live = model.clone()
        elif op == "stale" and E and rows.buf[rows.cur ^ 1] is not None and rows.buf[rows.cur ^ 1].shape[0] >= E:
            spare = rows.buf[rows.cur ^ 1]                            # ... or a view of the spare buffer
            spare[:E] = model
            live = spare[:E]
        assert live.shape == model.shape and torch.equal(live, model), (step, op)
    assert moved > 3


@pytest.mark.gpu
def test_native_updates_are_reproducible(cuda):
    """Repeated from the same state, a keyframe update (edge rebuild + 6 x pvo_graph_update on two streams) and a 64-keyframe
    global update (372 edges, envelope solve) give bit-identical poses, disparities, damping and hidden state: integer
    accumulation in the BA, fixed summation orders everywhere else, and no kernel on the side stream that a consumer does
    not wait for (SURVEY 8e asks for replicas that agree exactly)."""
    import bench
    video, graph = bench.make_window(cuda, seed=0)
    snap = bench.Snapshot(video, graph)
    snap.edge_list = list(zip(graph._ii_h, graph._jj_h))
    outs = []
    for r in range(8):
        bench.keyframe_update(video, graph, snap)
        torch.cuda.synchronize()
        outs.append((video.poses.clone(), video.disps.clone(), graph.damping.clone(), graph.net.clone()))
    for r in range(3, 8):                        # (the first steps settle the rebuilt edge order)
        for a, b in zip(outs[r], outs[2]):
            assert torch.equal(a, b), r
    nkf = 64
    video, graph = bench.make_window(cuda, seed=7, NKF=nkf, buffer=80, corr_impl="volume", add_edges=False, max_factors=-1)
    video.counter = nkf
    graph.add_factors([i for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 3],
                      [j for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 3])
    g = torch.Generator().manual_seed(3)
    graph.target_cam = graph.target_cam + 0.5 * torch.randn(graph.target_cam.shape, generator=g).to(cuda)
    state = [t.clone() for t in (video.poses, video.disps, graph.net, graph.target_cam, graph.damping)]
    outs = []
    for r in range(6):
        for dst, src in zip((video.poses, video.disps, graph.net, graph.target_cam, graph.damping), state):
            dst.copy_(src)
        graph.delta_dy.zero_(); graph.raw_mask.zero_()
        graph.update_lowmem(steps=2)
        torch.cuda.synchronize()
        outs.append((video.poses.clone(), video.disps.clone(), graph.damping.clone()))
    assert (outs[0][0] - state[0]).abs().max() > 1e-4
    for r in range(1, 6):
        for a, b in zip(outs[r], outs[0]):
            assert torch.equal(a, b), r


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_reproject_with_motion_features_equals_the_two_kernels(cuda, dtype):
    """pvo_reproject_motion (the head of pvo_graph_update) = pvo_reproject followed by pvo_graph_motion, bit for bit, on a map
    whose size is not a multiple of the workgroup (30 x 101) and with out-of-range targets (the +-64 clamp)"""
    from pvo_amd import droid_backends as db
    from pvo_amd.geom.se3 import SE3
    g = torch.Generator().manual_seed(5)
    F, ht, wd = 6, 30, 101
    poses = SE3.exp(0.05 * torch.randn(F, 6, generator=g)).data.to(cuda).contiguous()
    disps = (0.2 + torch.rand(F, ht, wd, generator=g)).to(cuda)
    intr = torch.tensor([90.0, 90.0, 50.5, 15.0]).repeat(F, 1).to(cuda)
    ii = torch.tensor([0, 1, 2, 3, 4, 5, 2, 0], device=cuda)
    jj = torch.tensor([1, 0, 4, 1, 2, 3, 5, 3], device=cuda)
    E = ii.numel()
    target = (200.0 * torch.randn(1, E, ht, wd, 2, generator=g)).to(cuda)
    ddy = torch.randn(1, E, ht, wd, 2, generator=g).to(cuda)
    raw = (40.0 * torch.randn(1, E, ht, wd, 2, generator=g)).to(cuda)
    c0, v0 = db.reproject(poses, disps, intr, ii, jj)
    m0 = db.graph_motion(target, c0[None].contiguous(), ddy, raw, dtype)
    c1, v1, m1 = db.reproject_motion(poses, disps, intr, ii, jj, target, ddy, raw, dtype)
    assert torch.equal(c0, c1) and torch.equal(v0, v1)
    assert torch.equal(m0.view(torch.int16), m1.view(torch.int16))
    assert (m1.float().abs() == 64).any()


@pytest.mark.gpu
def test_context_computed_ahead_inside_the_pose_solves_changes_nothing(cuda):
    """pvo_graph_update computes the NEXT update's gate context (a function of the hidden state and the weights) inside this
    update's two pose-solve dispatches and the next call uses it if nobody wrote `net` in between: two keyframe steps (12
    updates, the first of every step recomputes - the step restores `net`) end in the same bits as with the riders off (pvo_debug_config: no_riders)
    and with the riders off altogether.  The choice is read once per process: one process per setting."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r); import bench; dev = torch.device('cuda:0'); "
            "video, graph = bench.make_window(dev, seed=0); snap = bench.Snapshot(video, graph); "
            "snap.edge_list = list(zip(graph._ii_h, graph._jj_h)); "
            "[bench.keyframe_update(video, graph, snap) for _ in range(4)]; torch.cuda.synchronize(); "
            "torch.save([t.cpu() for t in (video.poses, video.disps, graph.damping, graph.net, graph.target_cam, graph.weight)], sys.argv[1])") % root
    outs = []
    prelude = ("import sys; sys.path.insert(0, %r); from pvo_amd import droid_backends as _dbk; "
               "[_dbk.debug_config(kv.split('=')[0], int(kv.split('=')[1])) for kv in sys.argv[2:]]; ") % root
    for knobs in ([], ["no_riders=1"], ["post_separate=1"]):       # (post_separate: pvo_graph_post as a launch of its own, not the gather's epilogue)
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            r = subprocess.run([sys.executable, "-c", prelude + code, f.name] + knobs, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:]
            outs.append(torch.load(f.name))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_packed_index_upload_keeps_parts_apart():
    """to_device_packed: several index tables in one staging buffer, each part its own dtype, 16-byte aligned, empty parts allowed
    (the CPU branch shares the packing arithmetic's contract: one tensor per part, same values)"""
    from pvo_amd.droid_backends import to_device_packed
    parts = [([5, 6, 7], torch.long), ([], torch.int32), ([1], torch.int32), (torch.tensor([9, 8]), torch.long), ([3] * 37, torch.int32)]
    out = to_device_packed(parts, "cpu")
    assert len(out) == len(parts)
    for (v, dt), t in zip(parts, out):
        assert t.dtype == dt and t.tolist() == (v.tolist() if isinstance(v, torch.Tensor) else list(v))


def test_pool_reserve_and_add_must_agree():
    """CorrVolumePool.add with a caller-uploaded slot tensor refuses one that does not belong to the last reserve()"""
    from pvo_amd.modules.corr import CorrVolumePool
    pool = CorrVolumePool.__new__(CorrVolumePool)
    pool.free, pool.slots, pool._reserved, pool.capacity = [3, 2, 1, 0], [], None, 4
    assert pool.reserve(2) == [0, 1] and pool.free == [3, 2]
    with pytest.raises(ValueError):
        pool.add(torch.zeros(3, 2, 2, 8), torch.zeros(3, 2, 2, 8), torch.zeros(2, dtype=torch.int32))      # 3 edges, 2 slots
    pool._reserved = [0, 1]
    with pytest.raises(ValueError):
        pool.add(torch.zeros(2, 2, 2, 8), torch.zeros(2, 2, 2, 8), torch.zeros(2, dtype=torch.int64))      # wrong dtype


def test_pinned_ring_exact_fit_and_wraparound():
    """_PinnedRing.take: an allocation that ends exactly on a quarter boundary (or on the end of the ring) must still
    record the event of the quarter being left, wait for the event of the quarter being entered, wrap, and never hand
    out an empty or overlapping segment while copies from it may be in flight (`off // quarter` named the
    NEXT quarter after an exact fit; 64 x 16 KB then take(64) raised, and every later upload failed)."""
    from pvo_amd.droid_backends import _PinnedRing

    class Ev:
        def __init__(self, log, q): self.log, self.q, self.done = log, q, False
        def synchronize(self): self.done = True; self.log.append(("sync", self.q))

    ring = _PinnedRing(nbytes=1 << 12, pin=False)          # quarters of 1024 B
    log = []
    ring._record = lambda device: (log.append(("record", ring.q)), Ev(log, ring.q))[1]
    base = ring.buf.data_ptr()
    live = {}                                              # quarter -> list of (lo, hi) handed out since its last wait
    import random
    rnd = random.Random(0)
    sizes = [1024] * 9 + [16] * 64 + [1024, 64, 960, 1024] + [rnd.choice([16, 48, 64, 512, 1008, 1024]) for _ in range(4000)]
    for n in sizes:
        before = len(log)
        seg = ring.take(n, "cpu")
        assert seg is not None and seg.numel() == ((n + 15) & ~15), n
        lo = seg.data_ptr() - base
        hi = lo + seg.numel()
        q = lo // 1024
        assert q == (hi - 1) // 1024 == ring.q and hi <= 4096        # inside one quarter, the one the ring says is current
        for kind, qq in log[before:]:
            if kind == "sync":
                live[qq] = []                                        # its copies have completed: reusable
        if any(k == "record" for k, _ in log[before:]):              # moved on: the entered quarter must be clean
            assert not live.get(q), "quarter %d reused without waiting for its copies" % q
        for a, b in live.setdefault(q, []):
            assert hi <= a or lo >= b, "overlapping segments in quarter %d" % q
        live[q].append((lo, hi))
    assert ring.take(1025, "cpu") is None                             # larger than a quarter: caller's one-off path
    recs = [q for k, q in log if k == "record"]
    assert all((b - a) % 4 == 1 for a, b in zip(recs, recs[1:]))      # quarters are left in order 0,1,2,3,0,...


@pytest.mark.gpu
def test_global_update_never_applies_the_panoptic_vote(cuda):
    """with segm_filter on, update_lowmem's resident-volume branch (one pvo_graph_update per step) applied the
    panoptic segment vote although the reference's global update never does (factor_graph.py:309-360).  Both branches on a
    video that filters by segments, from a mask for which the vote WOULD force whole segments: they must agree to the
    operator's fp16 rounding, weights and residual flow included."""
    import bench
    nkf = 8
    ii = [i for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 2]
    jj = [j for i in range(nkf) for j in range(nkf) if i != j and abs(i - j) <= 2]
    res = {}
    for name, corr_impl, fused in (("volume", "volume", True), ("alt", "alt", False)):
        video, graph = bench.make_window(cuda, seed=5, H8=16, W8=24, NKF=nkf, buffer=16, corr_impl=corr_impl, add_edges=False, max_factors=-1,
                                         intr=(15.0, 15.0, 12.0, 8.0))
        video.counter = nkf
        video.segm_filter, video.thresh, video.max_segments = True, 0.4, 16
        blocks = ((torch.arange(16)[:, None] // 4) * 3 + torch.arange(24)[None] // 8).int().to(cuda)          # ids 0..11, 0 = no segment
        video.segms[:nkf] = blocks[None, None]
        graph.fused_glue = fused
        graph.add_factors(list(ii), list(jj))
        g = torch.Generator().manual_seed(9)
        # half of the panoptic blocks are 70 % dynamic (above the 0.4 threshold): a vote would force their remaining 30 %
        start = (torch.rand(1, len(ii), 4, 3, generator=g) < 0.5).float()
        start = torch.nn.functional.interpolate(start, size=(16, 24))[..., None]
        start = (start * (torch.rand(1, len(ii), 16, 24, 1, generator=g) < 0.7).float()).to(cuda)
        graph.raw_mask = ((1.0 - 2.0 * start) * 0.3).expand(graph.raw_mask.shape).contiguous()
        b = torch.sigmoid(graph.raw_mask) >= graph.dy_thresh
        assert (graph._segment_vote(b) != b).sum().item() > 50           # the vote would change this mask
        graph.update_lowmem(steps=1)
        torch.cuda.synchronize()
        res[name] = dict(weight=graph.weight.clone(), delta_dy=graph.delta_dy.clone(), raw=graph.raw_mask.clone(), poses=video.poses.clone())
    far = (res["volume"]["raw"].abs() > 5e-3) & (res["alt"]["raw"].abs() > 5e-3)       # (pixels on the hard threshold may flip: excluded)
    for k in ("weight", "delta_dy"):
        dd = ((res["volume"][k] - res["alt"][k]).abs() * far)
        assert dd.max().item() < 2e-2 and dd.mean().item() < 1e-3, (k, dd.max().item(), dd.mean().item())
    assert (res["volume"]["poses"] - res["alt"]["poses"]).abs().max() < 1e-3


def test_proximity_select_host_function_argument_checks_and_corner_cases():
    """pvo_proximity_select called directly: nothing under the threshold leaves the temporal neighbours; no existing edges; a window whose
    two frame ranges do not end together and an output buffer that is too small are refused"""
    import ctypes
    import numpy as np
    from pvo_amd import droid_backends as db
    from pvo_amd import _lib
    d = np.full((6, 6), 50.0, dtype=np.float32)
    ii, jj = db.proximity_select(d, 0, 0, 2, 1, 10.0, np.zeros(0, np.int64), np.zeros(0, np.int64))
    want = []
    for i in range(6):
        for j in range(i + 1, min(i + 3, 6)):
            want += [(i, j), (j, i)]
    assert list(zip(ii, jj)) == want
    d[5, 1] = 3.0; d[4, 0] = 3.0                      # a tie: the lower flat index first; its diamond (radius min(|4 - 0| - 2, nms) = 2) holds (5, 1)
    ii, jj = db.proximity_select(d, 0, 0, 2, 2, 10.0, np.zeros(0, np.int64), np.zeros(0, np.int64))
    assert list(zip(ii, jj))[len(want):] == [(4, 0), (0, 4)]
    ii, jj = db.proximity_select(d, 0, 0, 2, 0, 10.0, np.zeros(0, np.int64), np.zeros(0, np.int64))     # nms 0: both survive
    assert list(zip(ii, jj))[len(want):] == [(4, 0), (0, 4), (5, 1), (1, 5)]
    ii, jj = db.proximity_select(d, 0, 0, 2, 0, 10.0, np.array([4], np.int64), np.array([0], np.int64))  # an existing edge takes its own cell out
    assert list(zip(ii, jj))[len(want):] == [(5, 1), (1, 5)]
    lib = _lib.load()
    out = np.zeros((2, 64), np.int64); n = ctypes.c_int(0)
    call = lambda ni, nj, t0, t1, cap: lib.pvo_proximity_select(d.ctypes.data, ni, nj, t0, t1, 2, 1, 10.0, None, None, 0, out[0].ctypes.data, out[1].ctypes.data, cap, ctypes.byref(n))
    assert call(6, 6, 0, 0, 64) == 0 and n.value == len(want) + 4     # (nms 1: both cells under the threshold are taken)
    assert call(6, 6, 0, 1, 64) != 0                  # t0 + ni != t1 + nj
    assert call(6, 6, 0, 0, 4) != 0                   # does not fit
