"""Training losses of the VO module — what VO_Module/train.py combines into its objective.

Counterpart of the reference's `geom/losses.py` (VO_Module/droid_slam/geom/losses.py).  Every loss is a
gamma-weighted sum over the unrolled update steps, `sum_i gamma^(n-1-i) * term_i` (gamma = 0.9), and returns
`(loss, metrics)` with the reference's metric names, so `tools/train.py` logs what `train.py:245-268` logs.

    residual_loss      :83-93     mean |BA residual|
    geodesic_loss      :31-80     SE3 log of the relative-pose error per edge (do_scale needs Sim3: not on PVO's path,
                                  train.py:150 passes do_scale=False)
    cam_flow_loss      :96-128    EPE of the camera-induced flow on the consecutive-frame graph
    flow_loss          :131-157   EPE against ground-truth forward / backward flow
    photo_loss         :160-223   photometric error of image j warped by the predicted full flow
    photo_loss_cam     :226-273   ... warped by the camera-induced flow (static pixels)
    gt_label_loss      :466-493   cross entropy of the static/dynamic mask
    art_label_loss, unsup_art_label, unsup_occ_vals, unsup_dy_vals  :276-344, :401-464  the unsupervised mode's
                                  artificial labels and occlusion masks
    ce_reg_loss, consistency_loss :382-398, :502-531
    SSIM               :363-393   3x3 average-pool structural dissimilarity, reflection padded

Everything stays on the device the inputs are on (the reference moves the unsupervised masks through the CPU).
"""
from collections import OrderedDict

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .graph_utils import graph_to_edge_list
from .projective_ops import coords_grid, projective_transform

GAMMA = 0.9


def _weighted(n, term, gamma=GAMMA):
    """sum_i gamma^(n-1-i) term(i); also returns the last term (the metrics report the final step)"""
    total, last = 0.0, None
    for i in range(n):
        last = term(i)
        total = total + gamma ** (n - 1 - i) * last
    return total, last


def _chain_graph(n):
    g = OrderedDict()
    for i in range(n):
        g[i] = [j for j in range(n) if abs(i - j) == 1]
    return g


def _edges(graph, device):
    ii, jj, _ = graph_to_edge_list(graph)
    return ii.to(device), jj.to(device)


def mean_on_mask(diff, val_pix):
    """mean of diff over the pixels val_pix marks; 0 (with the reference's warning) when fewer than 10^4 are marked"""
    m = val_pix.expand_as(diff)
    s = m.sum()
    if s > 10000:
        return (diff * m).sum() / s
    print("warning - most pixels are masked.")
    return torch.zeros((), dtype=m.dtype, device=m.device)


def ce_func(labels, inputs):
    """binary cross entropy with the reference's 1e-10 guards (losses.py:496-499)"""
    return -(labels * torch.log(inputs + 1e-10) + (1 - labels) * torch.log(1 - inputs + 1e-10))


class SSIM(nn.Module):
    """(1 - SSIM) / 2 over 3x3 windows, clamped to [0, 1] (losses.py:363-393)"""
    C1, C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        x, y = F.pad(x, (1, 1, 1, 1), mode="reflect"), F.pad(y, (1, 1, 1, 1), mode="reflect")
        pool = lambda t: F.avg_pool2d(t, 3, 1)
        mx, my = pool(x), pool(y)
        vx, vy, cxy = pool(x * x) - mx * mx, pool(y * y) - my * my, pool(x * y) - mx * my
        num = (2 * mx * my + self.C1) * (2 * cxy + self.C2)
        den = (mx * mx + my * my + self.C1) * (vx + vy + self.C2)
        return torch.clamp((1 - num / den) / 2, 0, 1)


def compute_reprojection_loss(pred, target, ssim):
    """per-pixel photometric error: L1 over channels, or 0.85 SSIM + 0.15 L1 (losses.py:347-360)"""
    l1 = (target - pred).abs().mean(1)
    if ssim is None:
        return l1
    return 0.85 * ssim(pred, target).mean(1) + 0.15 * l1


def residual_loss(residuals, gamma=GAMMA):
    loss, _ = _weighted(len(residuals), lambda i: residuals[i].abs().mean(), gamma)
    return loss, {"residual": loss.item()}


def _so3_angle(q):
    """rotation angle of a unit quaternion (xyzw): |log|"""
    v, w = q[..., :3].norm(dim=-1), q[..., 3]
    return 2 * torch.atan2(v, w.abs())


def geodesic_loss(Ps, Gs, graph, gamma=GAMMA, do_scale=False):
    """|tau| + |phi| of log(dG dP^-1) per edge, dX = X_j X_i^-1 (losses.py:31-80)"""
    if do_scale:
        raise NotImplementedError("do_scale fits a Sim3 scale; PVO's train.py passes do_scale=False (train.py:150)")
    ii, jj = _edges(graph, Ps.data.device)
    dP = Ps[:, jj] * Ps[:, ii].inv()
    err = {}

    def term(i):
        dG = Gs[i][:, jj] * Gs[i][:, ii].inv()
        dE = dG * dP.inv()
        d = dE.log()
        err["E"] = dE
        return d[..., :3].norm(dim=-1).mean() + d[..., 3:].norm(dim=-1).mean()
    loss, _ = _weighted(len(Gs), term, gamma)
    data = err["E"].data.detach()
    r_err = (180 / math.pi) * _so3_angle(data[..., 3:7])
    t_err = data[..., :3].norm(dim=-1)
    return loss, {"rot_error": r_err.mean().item(), "tr_error": t_err.mean().item(),
                  "bad_rot": (r_err < .1).float().mean().item(), "bad_tr": (t_err < .01).float().mean().item()}


def cam_flow_loss(Ps, disps, poses_est, disps_est, intrinsics, graph, gamma=GAMMA):
    """EPE between the flow induced by the estimated and by the true geometry, consecutive frames (losses.py:96-128; the
    graph argument is replaced by the chain graph, as there)"""
    ii, jj = _edges(_chain_graph(Ps.shape[1]), disps.device)
    coords0, val0 = projective_transform(Ps, disps, intrinsics, ii, jj)
    val0 = val0 * (disps[:, ii] > 0).float().unsqueeze(-1)
    keep = {}

    def term(i):
        coords1, val1 = projective_transform(poses_est[i], disps_est[i], intrinsics, ii, jj)
        v = (val0 * val1).squeeze(-1)
        keep["v"], keep["epe"] = v, v * (coords1 - coords0).norm(dim=-1)
        return keep["epe"].mean()
    loss, _ = _weighted(len(poses_est), term, gamma)
    epe = keep["epe"].reshape(-1)[keep["v"].reshape(-1) > 0.5]
    return loss, {"f_error": epe.mean().item(), "1px": (epe < 1.0).float().mean().item()}


def flow_loss(fo_flows, ba_flows, full_flows, graph, gamma=GAMMA):
    """forward edges are the even rows of the edge list, backward the odd ones (losses.py:131-157)"""
    def term(i):
        fo = ((full_flows[i][:, 0::2] - fo_flows[..., 0:2]).norm(dim=-1) * fo_flows[..., 2]).mean()
        ba = ((full_flows[i][:, 1::2] - ba_flows[..., 0:2]).norm(dim=-1) * ba_flows[..., 2]).mean()
        return (fo + ba) / 2
    loss, last = _weighted(len(full_flows), term, gamma)
    return loss, {"pure_f_error": last.item()}


def _warp(images1, coords, ht, wd):
    """sample images1 at pixel coordinates `coords` (align_corners grid, border padding); also the in-image mask"""
    grid = torch.stack([coords[..., 0] / (wd - 1), coords[..., 1] / (ht - 1)], dim=-1).view(-1, ht, wd, 2) * 2 - 1
    inside = (grid.abs().max(-1)[0] <= 1).float()
    return F.grid_sample(images1, grid, padding_mode="border", align_corners=True), inside


def photo_loss(images, full_flows, vals, graph, mode, gamma=GAMMA, ssim=None, mean_mask=False, aff_params=None,
               downsample=False):
    """photometric error of frame j warped to frame i by the predicted flow (losses.py:160-223)"""
    C = images.shape[2]
    ii, jj = _edges(graph, images.device)
    if downsample:
        images = images[..., 3::8, 3::8]
    ht, wd = images.shape[-2:]
    if mode != "unsup":
        vals_all = vals[..., 3::8, 3::8, :][:, ii].reshape(-1, ht, wd)
    im0 = images[:, ii].reshape(-1, C, ht, wd) / 255.0
    im1 = images[:, jj].reshape(-1, C, ht, wd) / 255.0
    coords0 = coords_grid(ht, wd, device=images.device)
    keep = {}

    def term(i):
        warped, inside = _warp(im1, coords0 + full_flows[i], ht, wd)
        v = vals[i].to(images.device).view(-1, ht, wd) if mode == "unsup" else vals_all
        val_pix = inside * v
        if aff_params is not None:
            warped = warped * aff_params[i][..., 0].view(-1, 1, 1, 1) + (aff_params[i][..., 1] - 0.5).view(-1, 1, 1, 1)
        diff = compute_reprojection_loss(im0, warped, ssim)
        keep["diff"], keep["val"] = diff, val_pix
        return mean_on_mask(diff, val_pix) if mean_mask else (diff * val_pix).mean()
    loss, last = _weighted(len(full_flows), term, gamma)
    return loss, {"ph_error": last.item(), "0.01color": mean_on_mask((keep["diff"] < 0.01).float(), keep["val"]).item()}


def photo_loss_cam(images, poses_est, disps_est, intrinsics, graph, mode, masks, gamma=GAMMA, ssim=None):
    """photometric error under the camera-induced flow on consecutive frames, static pixels only (losses.py:226-273)"""
    C = images.shape[2]
    ht, wd = images.shape[-2:]
    ii, jj = _edges(_chain_graph(images.shape[1]), images.device)
    im0 = images[:, ii].reshape(-1, C, ht, wd) / 255.0
    im1 = images[:, jj].reshape(-1, C, ht, wd) / 255.0
    if mode != "unsup":
        masks_all = masks[:, ii].reshape(-1, ht, wd)
    keep = {}

    def term(i):
        coords, val0 = projective_transform(poses_est[i], disps_est[i], intrinsics, ii, jj)
        warped, inside = _warp(im1, coords, ht, wd)
        m = masks[i].to(images.device).view(-1, ht, wd) if mode == "unsup" else masks_all
        val_pix = inside * val0.view(-1, ht, wd) * m
        diff = compute_reprojection_loss(im0, warped, ssim)
        keep["diff"], keep["val"] = diff, val_pix
        return (diff * val_pix).mean()
    loss, last = _weighted(len(poses_est), term, gamma)
    return loss, {"ph_cam_error": last.item(),
                  "0.01color_cam": mean_on_mask((keep["diff"] < 0.01).float(), keep["val"]).item()}


def gt_label_loss(gt_masks, gt_vals, masks, graph, gamma=GAMMA, mean_mask=False):
    """cross entropy of the predicted static probability against the ground-truth mask of the source frame (losses.py:466-493)"""
    ii, _ = _edges(graph, gt_masks.device)
    lab, val = gt_masks[:, ii], gt_vals[:, ii]

    def term(i):
        diff = ce_func(lab, masks[i])
        return mean_on_mask(diff, val) if mean_mask else (diff * val).mean()
    loss, last = _weighted(len(masks), term, gamma)
    return loss, {"gt_mask_error": last.item(), "static_px_rate": (lab * val).mean().item(),
                  "dynamic_px_rate": ((1 - lab) * val).mean().item()}


def ce_reg_loss(preds, gamma=GAMMA):
    loss, last = _weighted(len(preds), lambda i: (-preds[i] * torch.log(preds[i] + 1e-10)).sum(-1).mean(), gamma)
    return loss, {"mask_entro_error": last.item()}


def consistency_loss(masks, n_frames, graph, gamma=GAMMA):
    """masks of the edges leaving one frame should agree (losses.py:502-531; as there the deviation is averaged signed, and the
    per-frame edge ranges are the reference's - see below; train.py leaves this loss off by default)"""
    ii, _, _ = graph_to_edge_list(graph)
    start = [0] * (n_frames + 1)
    for i in ii.tolist():
        start[i + 1] += 1
    for i in ii.tolist():                 # (once per EDGE, not per frame, as the reference accumulates them, losses.py:509-510:
        start[i + 1] += start[i]          #  with several edges per frame these are not the prefix sums - slices can be empty)

    def term(i):
        e = 0.0
        for f in range(n_frames):
            m = masks[i][:, start[f]:start[f + 1]]
            e = e + (m - m.mean(1, keepdim=True)).mean()
        return e / n_frames
    loss, last = _weighted(len(masks), term, gamma)
    return loss, {"con_error": last.item()}


def _upsample8(x):
    from ..droid_net import upsample_inter
    return upsample_inter(x)


def unsup_art_label(poses_est, disps_est, intrinsics, full_flows, graph, thresh=0.5, downsample=True):
    """artificial static labels: pixels whose predicted flow agrees with the camera-induced flow within `thresh` px
    (losses.py:401-432).  `intrinsics` is not modified (the reference divides its CPU copy in place)."""
    ht, wd = full_flows[0].shape[2:4]
    dev = full_flows[0].device
    ii, jj = _edges(graph, dev)
    intr = intrinsics / 8 if downsample else intrinsics
    coords0 = coords_grid(ht, wd, device=dev)
    out = []
    for flow, G, d in zip(full_flows, poses_est, disps_est):
        d = d.detach()[:, :, 3::8, 3::8] if downsample else d.detach()
        cam, _ = projective_transform(G.detach(), d, intr, ii, jj)
        out.append(((coords0 + flow.detach() - cam).norm(dim=-1) <= thresh).float().unsqueeze(-1))
    return out


def art_label_loss(art_masks, masks, gamma=GAMMA, downsample=True):
    keep = {}

    def term(i):
        keep["m"] = _upsample8(art_masks[i]) if downsample else art_masks[i]
        return ce_func(keep["m"].to(masks[i].device), masks[i]).mean()
    loss, last = _weighted(len(masks), term, gamma)
    rate = keep["m"].mean().item()
    return loss, {"art_mask_error": last.item(), "static_px_rate": rate, "dynamic_px_rate": 1 - rate}


def unsup_occ_vals(poses_est, disps_est, intrinsics, downsample, graph, loss, use_one=False):
    """occlusion masks from depth consistency between frame i's points seen from j and frame j's own depth
    (losses.py:276-321)"""
    from .projective_ops import projective_transform_unsup
    N = disps_est[0].shape[1]
    dev = disps_est[0].device
    ii, jj = _edges(graph if graph is not None else _chain_graph(N), dev)
    intr = intrinsics / 8 if downsample else intrinsics
    out = []
    for G, d in zip(poses_est, disps_est):
        d = d.detach()[:, :, 3::8, 3::8] if downsample else d.detach()
        ht, wd = d.shape[2:]
        if use_one:
            out.append(torch.ones_like(d[:, jj].reshape(-1, 1, ht, wd)))
            continue
        cam, disp0, _ = projective_transform_unsup(G.detach(), d, intr, ii, jj)
        disp0 = disp0.reshape(-1, 1, ht, wd)
        warped, _ = _warp(d[:, jj].reshape(-1, 1, ht, wd), cam, ht, wd)
        if loss == "ph_loss":
            out.append(((1 / warped - 1 / disp0) > -0.005).float())
        else:
            out.append(((1 / disp0 - 1 / warped).abs() <= 0.005).float())
    return out


def unsup_dy_vals(vals, dy_masks, graph):
    """a pixel stays valid if it is unoccluded or dynamic (losses.py:324-344)"""
    ii, _, _ = graph_to_edge_list(graph)
    fixed = None
    if not isinstance(dy_masks, list):
        m = dy_masks.detach()[:, :, 3::8, 3::8]
        fixed = m[:, ii.to(m.device)].reshape(-1, 1, *m.shape[2:4])
    out = []
    for i, v in enumerate(vals):
        if fixed is not None:
            fixed = 1 - fixed             # (the reference complements its one mask tensor again in every step, losses.py:340: steps
            m = fixed                     #  0, 2, 4 ... see the complement, steps 1, 3 ... the mask itself)
        else:
            m = 1 - dy_masks[i].reshape(-1, 1, *dy_masks[i].shape[2:4])
        out.append(torch.clamp(v + m.to(v.device), min=0, max=1))
    return out
