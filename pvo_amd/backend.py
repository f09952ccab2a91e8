"""DroidBackend - global bundle adjustment over all keyframes (VO_Module/droid_slam/droid_backend.py:9-41).

A fresh factor graph with `corr_impl="alt"` (no stored volumes: features are correlated on the fly by the
alt-corr HIP kernel), proximity edges over the whole video, `update_lowmem` for `steps` iterations.
"""
import torch

from .factor_graph import FactorGraph


class DroidBackend:
    def __init__(self, net, video, args):
        self.video, self.update_op, self.device = video, net.update, args.device
        self.t0 = self.t1 = 0
        self.beta = args.beta
        self.backend_thresh, self.backend_radius, self.backend_nms = \
            args.backend_thresh, args.backend_radius, args.backend_nms

    @torch.no_grad()
    def __call__(self, steps=12):
        t = self.video.counter
        self.video.normalize()
        graph = FactorGraph(self.video, self.update_op, self.device, corr_impl="alt", max_factors=100000)
        graph.add_proximity_factors(rad=self.backend_radius, nms=self.backend_nms, thresh=self.backend_thresh,
                                    beta=self.beta)
        graph.update_lowmem(steps=steps)
        graph.clear_edges()
        self.video.dirty[:t] = True
