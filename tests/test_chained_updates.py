"""N chained graph updates: the native path (one pvo_graph_update per update: fp16 volume, fp16 operator, HIP BA) against the
CPU fp32 chain (oracle lookup -> fp32 operator -> oracle BA, oracle/chain.py) from the same S-B state.  VERDICT r2 item 8:
single updates were compared product-vs-product only; nothing bounded the drift over a keyframe's six updates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def structured_operator(op, scale=0.1):
    """Random-init heads emit O(1)-pixel noise as flow revisions, which the BA turns into depth noise and every later update
    amplifies (no trained weights exist here).  Scaling the LAST layer of the two flow heads keeps the revisions sub-pixel and
    spatially smooth-ish - the regime a trained operator works in near convergence - without touching any other arithmetic."""
    with torch.no_grad():
        for head in (op.delta, op.delta_dy):
            head[2].weight.mul_(scale); head[2].bias.mul_(scale)
    return op


def chained_drift(device, n_updates=6, seed=0, scale=0.1, segments=False, mask_seed=None):
    import bench
    from oracle import chain
    video, graph = bench.make_window(device, seed=seed, segments=segments, thresh=0.6)
    structured_operator(graph.update_op, scale)
    if mask_seed is not None:
        # start from a dynamic mask that covers some panoptic instances mostly, others barely: the vote (fraction > thresh)
        # has segments to force on some edges and segments to leave alone on others
        g = torch.Generator().manual_seed(mask_seed)
        E, H, W = graph.raw_mask.shape[1:4]
        seg = graph.segm[0, :, 0].long().cpu()                                            # [E,H,W]
        frac = torch.rand(E, 16, generator=g)                                             # per (edge, instance) dynamic fraction
        dyn = torch.rand(E, H, W, generator=g) < torch.gather(frac, 1, seg.view(E, -1)).view(E, H, W)
        raw = torch.where(dyn & (seg > 0), 0.6, -0.6) + 0.1 * torch.randn(E, H, W, generator=g)
        graph.raw_mask = raw[None, ..., None].expand(1, E, H, W, graph.raw_mask.shape[-1]).contiguous().to(device)
    ov, cg = chain.cpu_twin(video, graph, graph.nkf)
    p0, d0 = video.poses[:graph.nkf].clone(), video.disps[:graph.nkf].clone()
    rows = []
    for k in range(n_updates):
        graph.update(None, None, use_inactive=True)
        cg.update(None, None, use_inactive=True)
        torch.cuda.synchronize()
        n = graph.nkf
        forced = None
        if segments:
            # what the vote did on the CPU chain's state, and whether the device's mask agrees with the CPU chain's
            b = torch.sigmoid(cg.raw_mask) >= cg.dy_thresh
            forced = int((cg._segment_vote(b) != b).sum())
        rows.append(dict(
            forced=forced,
            mask_flips=float(((torch.sigmoid(graph.raw_mask.cpu()) >= 0.5) != (torch.sigmoid(cg.raw_mask) >= 0.5)).float().mean()),
            pose=float((video.poses[:n].cpu() - ov.poses[:n]).abs().max()),
            disp_mean=float((video.disps[:n].cpu() - ov.disps[:n]).abs().mean()), disp_max=float((video.disps[:n].cpu() - ov.disps[:n]).abs().max()),
            flow_epe=float((graph.target_cam.cpu() - cg.target_cam).norm(dim=-1).mean()),
            net=float((graph.net.float().cpu() - cg.net).abs().mean()),
            weight=float((graph.weight.cpu() - cg.weight).abs().mean())))
    moved = dict(pose=float((ov.poses[:graph.nkf] - p0.cpu()).abs().max()), disp=float((ov.disps[:graph.nkf] - d0.cpu()).abs().mean()))
    return rows, moved


def _show(rows, moved):
    for k, r in enumerate(rows):
        print("update %d: pose %.2e  disp mean %.2e max %.2e  flow EPE %.2e  net %.2e  weight %.2e  mask flips %.2e  forced %s" % (
            k + 1, r["pose"], r["disp_mean"], r["disp_max"], r["flow_epe"], r["net"], r["weight"], r["mask_flips"], r["forced"]))
    print("the chain moved poses by %.3e and depths by %.3e (mean)" % (moved["pose"], moved["disp"]))


def test_six_chained_native_updates_stay_within_fp16_drift_of_the_fp32_cpu_chain(cuda):
    """sub-pixel flow revisions (last layer of the two flow heads x 0.1): pose / depth / flow after each of six chained
    updates against the CPU fp32 chain - inside the north star's 1e-4 for poses and flow EPE at every step"""
    rows, moved = chained_drift(cuda, scale=0.1)
    _show(rows, moved)
    assert moved["pose"] > 1e-3 and moved["disp"] > 1e-3              # the six updates did move the state
    # observed on MI355X (round 4): pose <= 3.0e-7, flow EPE <= 6.2e-6 px, mean depth gap <= 1.44e-6, net <= 8.1e-5: bounds = 5 x that
    for r in rows:
        assert r["pose"] < 1.5e-6 and r["flow_epe"] < 3e-5 and r["disp_mean"] < 7e-6 and r["net"] < 4e-4, r
    last = rows[-1]
    assert last["pose"] < 5e-4 * moved["pose"] and last["disp_mean"] < 1.6e-3 * moved["disp"]
    assert last["disp_mean"] < 6 * max(rows[0]["disp_mean"], 3e-7)      # grows at most linearly with the number of updates


def test_chained_updates_with_the_unscaled_random_operator(cuda):
    """the same chain with the random-init heads as they are (O(1) px flow noise that the BA turns into depth noise): the
    regime round 2 compared product-vs-product after ONE update with poses < 1e-3, disps < 2e-2; here against the CPU fp32
    chain over six updates, as typical (mean) gaps - single pixels flip the hard mask threshold"""
    rows, moved = chained_drift(cuda, scale=1.0)
    _show(rows, moved)
    # observed on MI355X (round 4): first update pose 1.15e-6, flow EPE 4.3e-5 px, depth 3.8e-6; sixth 1.8e-6 / 4.9e-5 / 1.16e-5:
    # bounds = 5 x that (round 3 asserted 2e-3 / 2e-2 / 5e-3, four hundred times looser than what is observed)
    assert rows[0]["pose"] < 6e-6 and rows[0]["flow_epe"] < 2.2e-4 and rows[0]["disp_mean"] < 2e-5, rows[0]
    last = rows[-1]
    assert last["pose"] < 1e-5 and last["flow_epe"] < 2.5e-4 and last["disp_mean"] < 6e-5, last
    assert max(r["flow_epe"] for r in rows) < 1e-4                 # the north star's flow tolerance, at every step


def test_six_chained_native_updates_with_the_panoptic_vote_against_the_cpu_chain(cuda):
    """S-3 (BASELINE.json configs[2]): S-B + panoptic segments, segm_filter on.  The native update votes on the device
    (pvo_segment_hist + the test inside pvo_graph_post); the CPU chain votes in the PyTorch glue that the reference's
    FactorGraph.update pins (tests/golden/factor_graph_glue_segm.npz).  Same bounds as the chain without segments."""
    rows, moved = chained_drift(cuda, scale=0.1, segments=True, mask_seed=11)
    _show(rows, moved)
    assert moved["pose"] > 1e-3 and moved["disp"] > 1e-3
    assert rows[0]["forced"] > 500 and min(r["forced"] for r in rows) > 0         # the vote changed the mask in every update
    # observed (round 4): pose <= 4.8e-8, flow EPE <= 5.7e-6, mean depth gap <= 1.4e-6, mask flips <= 6.8e-5 of the pixels
    for r in rows:
        assert r["pose"] < 1.5e-6 and r["flow_epe"] < 3e-5 and r["disp_mean"] < 7e-6 and r["net"] < 4e-4, r
        assert r["mask_flips"] < 3.5e-4, r
