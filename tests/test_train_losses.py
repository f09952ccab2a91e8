"""Training losses (pvo_amd/geom/losses.py) against the reference's geom/losses.py, executed by tests/golden/gen_golden.py
::gen_losses on a seeded 4-frame clip (train_losses.npz holds only the reference's outputs; the inputs are regenerated
from the seed).  CPU: these are PyTorch formulations, the device path is the same code."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def case():
    import gen_golden as G
    return G.losses_case(), np.load(os.path.join(HERE, "golden", "train_losses.npz"))


def _check(ref, name, res, tol=2e-5):
    loss, metrics = res
    if np.isnan(float(ref[name])):                                   # (consistency_loss on a graph with several edges per frame: the
        assert np.isnan(float(loss)), name                           #  reference's edge ranges are not prefix sums, losses.py:509-510)
        return
    assert abs(float(loss) - float(ref[name])) <= tol * max(1.0, abs(float(ref[name]))), (name, float(loss), float(ref[name]))
    for k, v in metrics.items():
        want = float(ref[name + "/" + k])
        assert abs(v - want) <= 2e-4 * max(1.0, abs(want)), (name, k, v, want)
    assert {k.split("/", 1)[1] for k in ref.files if k.startswith(name + "/")} == set(metrics)       # the reference's metric names


def test_supervised_and_semisupervised_losses_match_the_reference(case):
    from pvo_amd.geom import losses as L
    c, ref = case
    ssim = L.SSIM()
    _check(ref, "residual", L.residual_loss(c["residuals"]))
    _check(ref, "geodesic", L.geodesic_loss(c["Ps"], c["poses_est"], c["graph"], do_scale=False))
    _check(ref, "cam_flow", L.cam_flow_loss(c["Ps"], c["disps"], c["poses_est"], c["disps_est"], c["intr"], c["graph"]))
    _check(ref, "flow", L.flow_loss(c["fo"], c["bo"], c["full_flows"], c["graph"]))
    _check(ref, "photo_sup_ds", L.photo_loss(c["images"], c["full_flows"], c["gt_vals"], c["graph"], "semisup", downsample=True))
    _check(ref, "photo_aff_ssim", L.photo_loss(c["images"], c["full_flows"], c["gt_vals"], c["graph"], "sup", ssim=ssim,
                                               aff_params=c["aff"], downsample=True, mean_mask=True))
    _check(ref, "photo_cam", L.photo_loss_cam(c["images"], c["poses_est"], c["disps_est"], c["intr"], c["graph"], "semisup",
                                              c["gt_masks"], ssim=ssim))
    _check(ref, "gt_label", L.gt_label_loss(c["gt_masks"], c["gt_vals"], c["masks"], c["graph"]))
    _check(ref, "gt_label_mean_mask", L.gt_label_loss(c["gt_masks"], c["gt_vals"], c["masks"], c["graph"], mean_mask=True))
    _check(ref, "ce_reg", L.ce_reg_loss(c["masks"]))
    _check(ref, "consistency", L.consistency_loss(c["masks"], c["N"], c["graph"]))
    x, y = c["images"][0, :2] / 255.0, c["images"][0, 1:3] / 255.0
    assert np.allclose(ssim(x, y).numpy(), ref["ssim_map"], atol=2e-6)
    with pytest.raises(NotImplementedError):
        L.geodesic_loss(c["Ps"], c["poses_est"], c["graph"], do_scale=True)


def test_unsupervised_labels_and_occlusion_masks_match_the_reference(case):
    from pvo_amd.geom import losses as L
    c, ref = case
    intr = c["intr"].clone()
    art = L.unsup_art_label(c["poses_est"], c["disps_est"], intr, c["full_flows"], c["graph"], downsample=True)
    assert torch.equal(intr, c["intr"])                               # (the reference divides its CPU copy in place; here the input is left alone)
    for k, a in enumerate(art):
        # a threshold on a float distance: allow the handful of pixels that sit on it
        assert (a.numpy() != ref["art_label_%d" % k]).mean() < 2e-3
    _check(ref, "art_label", L.art_label_loss([torch.from_numpy(ref["art_label_%d" % k]) for k in range(len(art))], c["masks"], downsample=True))
    for tag in ("ph_loss", "cam_ph_loss"):
        ds = tag == "ph_loss"
        vals = L.unsup_occ_vals(c["poses_est"], c["disps_est"], c["intr"], ds, c["graph"] if ds else None, tag)
        for k, v in enumerate(vals):
            want = ref["occ_%s_%d" % (tag, k)]
            assert v.shape == want.shape and (v.numpy() != want).mean() < 2e-3, (tag, k)
        if ds:
            dy = L.unsup_dy_vals([torch.from_numpy(ref["occ_ph_loss_%d" % k]) for k in range(len(vals))], c["gt_masks"][..., 0], c["graph"])
            for k, v in enumerate(dy):
                assert np.array_equal(v.numpy(), ref["dy_%d" % k])
    ones = L.unsup_occ_vals(c["poses_est"], c["disps_est"], c["intr"], True, c["graph"], "ph_loss", use_one=True)
    assert all(bool((o == 1).all()) for o in ones)


def test_losses_are_differentiable_through_the_network_outputs(case):
    """what train.py backpropagates: gradients reach residuals, flows, masks, poses and depths"""
    from pvo_amd.geom import losses as L
    from pvo_amd.geom.se3 import SE3
    c, _ = case
    res = [r.clone().requires_grad_() for r in c["residuals"]]
    flows = [f.clone().requires_grad_() for f in c["full_flows"]]
    masks = [m.clone().requires_grad_() for m in c["masks"]]
    pd = [G.data.clone().requires_grad_() for G in c["poses_est"]]
    de = [d.clone().requires_grad_() for d in c["disps_est"]]
    loss = (0.01 * L.residual_loss(res)[0] + 5.0 * L.photo_loss(c["images"], flows, c["gt_vals"], c["graph"], "semisup", downsample=True)[0]
            + 0.01 * L.gt_label_loss(c["gt_masks"], c["gt_vals"], masks, c["graph"])[0]
            + 100.0 * L.photo_loss_cam(c["images"], [SE3(p) for p in pd], de, c["intr"], c["graph"], "semisup", c["gt_masks"], ssim=L.SSIM())[0])
    loss.backward()
    for group in (res, flows, masks, pd, de):
        assert all(t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0 for t in group)


def test_training_frame_graph_matches_the_reference():
    """build_frame_graph / compute_distance_matrix_flow (geom/graph_utils.py:37-68, data_readers/rgbd_utils.py:110-152) on
    frame_graph_case(): the flow-distance matrix and the chosen edges, in order, for three (num, thresh) settings and both
    pose conventions"""
    import gen_golden as G
    from pvo_amd.geom.graph_utils import build_frame_graph, compute_distance_matrix_flow, graph_to_edge_list
    ref = np.load(os.path.join(HERE, "golden", "frame_graph.npz"))
    poses, disps, intr = G.frame_graph_case()
    for need_inv in (False, True):
        d = compute_distance_matrix_flow(poses[0], disps[0][:, 3::8, 3::8], intr[0] / 8.0, need_inv).numpy()
        want = ref["dist_inv%d" % need_inv]
        assert np.array_equal(np.isinf(d), np.isinf(want))
        fin = np.isfinite(want)
        assert np.allclose(d[fin], want[fin], rtol=1e-4, atol=1e-4)
        for num, thresh in ((20, 24.0), (40, 24.0), (30, 6.0)):
            g = build_frame_graph(poses, disps, intr, num=num, thresh=thresh, need_inv=need_inv)
            ii, jj, _ = graph_to_edge_list(g)
            assert np.array_equal(torch.stack([ii, jj]).numpy(), ref["edges_inv%d_%d_%g" % (need_inv, num, thresh)]), (need_inv, num, thresh)
