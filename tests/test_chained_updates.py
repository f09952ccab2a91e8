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


def chained_drift(device, n_updates=6, seed=0, scale=0.1):
    import bench
    from oracle import chain
    video, graph = bench.make_window(device, seed=seed)
    structured_operator(graph.update_op, scale)
    ov, cg = chain.cpu_twin(video, graph, graph.nkf)
    p0, d0 = video.poses[:graph.nkf].clone(), video.disps[:graph.nkf].clone()
    rows = []
    for k in range(n_updates):
        graph.update(None, None, use_inactive=True)
        cg.update(None, None, use_inactive=True)
        torch.cuda.synchronize()
        n = graph.nkf
        rows.append(dict(
            pose=float((video.poses[:n].cpu() - ov.poses[:n]).abs().max()),
            disp_mean=float((video.disps[:n].cpu() - ov.disps[:n]).abs().mean()), disp_max=float((video.disps[:n].cpu() - ov.disps[:n]).abs().max()),
            flow_epe=float((graph.target_cam.cpu() - cg.target_cam).norm(dim=-1).mean()),
            net=float((graph.net.float().cpu() - cg.net).abs().mean()),
            weight=float((graph.weight.cpu() - cg.weight).abs().mean())))
    moved = dict(pose=float((ov.poses[:graph.nkf] - p0.cpu()).abs().max()), disp=float((ov.disps[:graph.nkf] - d0.cpu()).abs().mean()))
    return rows, moved


def _show(rows, moved):
    for k, r in enumerate(rows):
        print("update %d: pose %.2e  disp mean %.2e max %.2e  flow EPE %.2e  net %.2e  weight %.2e" % (
            k + 1, r["pose"], r["disp_mean"], r["disp_max"], r["flow_epe"], r["net"], r["weight"]))
    print("the chain moved poses by %.3e and depths by %.3e (mean)" % (moved["pose"], moved["disp"]))


def test_six_chained_native_updates_stay_within_fp16_drift_of_the_fp32_cpu_chain(cuda):
    """sub-pixel flow revisions (last layer of the two flow heads x 0.1): pose / depth / flow after each of six chained
    updates against the CPU fp32 chain - inside the north star's 1e-4 for poses and flow EPE at every step"""
    rows, moved = chained_drift(cuda, scale=0.1)
    _show(rows, moved)
    assert moved["pose"] > 1e-3 and moved["disp"] > 1e-3              # the six updates did move the state
    for r in rows:
        assert r["pose"] < 1e-5 and r["flow_epe"] < 1e-4 and r["disp_mean"] < 2e-5 and r["net"] < 5e-4, r
    last = rows[-1]
    assert last["pose"] < 1e-3 * moved["pose"] and last["disp_mean"] < 2e-3 * moved["disp"]
    assert last["disp_mean"] < 6 * max(rows[0]["disp_mean"], 3e-7)      # grows at most linearly with the number of updates


def test_chained_updates_with_the_unscaled_random_operator(cuda):
    """the same chain with the random-init heads as they are (O(1) px flow noise that the BA turns into depth noise): the
    regime round 2 compared product-vs-product after ONE update with poses < 1e-3, disps < 2e-2; here against the CPU fp32
    chain over six updates, as typical (mean) gaps - single pixels flip the hard mask threshold"""
    rows, moved = chained_drift(cuda, scale=1.0)
    _show(rows, moved)
    assert rows[0]["pose"] < 1e-4 and rows[0]["flow_epe"] < 2e-3 and rows[0]["disp_mean"] < 2e-4, rows[0]
    last = rows[-1]
    assert last["pose"] < 2e-3 and last["flow_epe"] < 2e-2 and last["disp_mean"] < 5e-3, last
