"""TEST INFRASTRUCTURE (imported only by tests/ and bench.py's cpu_baseline leg): the graph update as a CPU fp32 chain.

    oracle lookup of an fp32 volume  ->  the update operator in fp32 (PyTorch, CPU)  ->  the oracle's dense BA (C)

driven by the product's own host logic (`pvo_amd.factor_graph.FactorGraph` on device "cpu", whose PyTorch glue is pinned
against the reference's FactorGraph.update by tests/golden/factor_graph_glue_*.npz).  It is what N chained native updates
(fp16 volume + fp16 operator + HIP BA on the GPU) are compared with, and - timed - the CPU port of the whole hot path.
"""
import copy

import numpy as np
import torch

from . import oracle as O


class OracleVideo:
    """DepthVideo's interface on CPU tensors; reproject / distance / ba answered by the C oracle"""

    def __init__(self, ht8, wd8, buffer):
        self.ht, self.wd, self.counter = ht8 * 8, wd8 * 8, 0
        self.poses = torch.zeros(buffer, 7); self.poses[:, 6] = 1
        self.disps = torch.ones(buffer, ht8, wd8)
        self.intrinsics = torch.zeros(buffer, 4)
        self.tstamp = torch.zeros(buffer); self.dirty = torch.zeros(buffer, dtype=torch.bool)
        self.segms = torch.zeros(buffer, 1, ht8, wd8, dtype=torch.int)
        self.segm_filter, self.thresh, self.max_segments = False, 0.8, 1024
        self.nets = self.inps = self.fmaps = None

    def reproject(self, ii, jj):
        c, v = O.reproject(self.poses.numpy(), self.disps.numpy(), self.intrinsics.numpy(), np.asarray(ii), np.asarray(jj))
        return torch.from_numpy(c)[None], torch.from_numpy(v)[None]

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        r = O.ba(self.poses.numpy(), self.disps.numpy(), self.intrinsics[0].numpy(), target.numpy(), weight.numpy(),
                 eta.numpy(), ii.numpy(), jj.numpy(), t0, t1, itrs, lm, ep, motion_only=motion_only)
        self.poses.copy_(torch.from_numpy(r["poses"])); self.disps.copy_(torch.from_numpy(r["disps"]).clamp(min=0.001))


class OracleCorr:
    """CorrBlock's interface: fp32 all-pairs volume + pyramid by torch.matmul / avg_pool2d (the reference's own formulation,
    modules/corr.py:24-38,63-71, pinned by tests/golden/corr_volume.npz), sampled by the C oracle's lookup"""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, channels_last=False):
        from pvo_amd.modules.corr import CorrBlock
        if channels_last:
            fmap1, fmap2 = fmap1.permute(0, 1, 4, 2, 3), fmap2.permute(0, 1, 4, 2, 3)
        self.radius = radius
        self.pyr = [p.contiguous().numpy() for p in CorrBlock._build_differentiable(fmap1.float(), fmap2.float(), num_levels)]

    def cat(self, other):
        self.pyr = [np.concatenate([a, b], 0) for a, b in zip(self.pyr, other.pyr)]
        return self

    def __call__(self, coords, **kw):
        b, n, h, w, _ = coords.shape
        return torch.from_numpy(O.corr_pyramid_lookup(self.pyr, coords.reshape(b * n, h, w, 2).numpy(), self.radius)).view(b, n, -1, h, w)


def cpu_twin(video, graph, n_frames):
    """a CPU fp32 copy of a device window: same poses / depths / features / per-edge state / edge order, the operator's
    (16-bit rounded) weights upcast to fp32"""
    import pvo_amd.factor_graph as FG
    ht8, wd8 = graph.ht, graph.wd
    ov = OracleVideo(ht8, wd8, video.poses.shape[0])
    ov.counter = n_frames
    ov.poses.copy_(video.poses.cpu()); ov.disps.copy_(video.disps.cpu()); ov.intrinsics.copy_(video.intrinsics.cpu())
    ov.fmaps, ov.nets, ov.inps = video.fmaps.float().cpu(), video.nets.float().cpu(), video.inps.float().cpu()
    # panoptic vote (factor_graph.py:256-276): the twin votes with the same dense labels, threshold and histogram width
    ov.segms.copy_(video.segms.cpu())
    ov.segm_filter, ov.thresh, ov.max_segments = bool(video.segm_filter), float(video.thresh), int(video.max_segments)
    op = copy.deepcopy(graph.update_op).float().cpu().eval()
    old = FG.CorrBlock
    FG.CorrBlock = OracleCorr
    try:
        g = FG.FactorGraph(ov, op, device="cpu", corr_impl="volume", max_factors=graph.max_factors)
        g.add_factors(list(graph._ii_h), list(graph._jj_h))
    finally:
        FG.CorrBlock = old
    assert g._ii_h == graph._ii_h and g._jj_h == graph._jj_h
    for name in ("target_cam", "weight", "raw_mask", "delta_dy"):
        setattr(g, name, getattr(graph, name).detach().float().cpu().clone())
    g.net = graph.net.detach().float().cpu().contiguous().clone()
    g.damping.copy_(graph.damping.cpu())
    return ov, g
