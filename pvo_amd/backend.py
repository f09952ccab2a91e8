"""Global bundle adjustment over every keyframe collected so far.

Role of the reference's `DroidBackend` (VO_Module/droid_slam/droid_backend.py:9-41), same constructor and call
signature.  One pass = rescale the map to unit mean inverse depth, connect all keyframes by proximity, iterate
`FactorGraph.update_lowmem`, drop the edges again.

Correlation features: `args.backend_corr = "alt"` (default, the reference's path: no stored volumes, features correlated
on the fly by the alt-corr HIP kernel, operator in 8-frame chunks) or `"volume"`: on MI355X the volumes of every edge of
the global graph fit in HBM (25 MB per edge of 288 GB), so the global update runs like the frontend's - resident tiled
pool, one native call per step for the whole graph (~8x faster at 372 edges; values carry the volume's fp16 rounding).
"""
import torch

from .factor_graph import FactorGraph

_EDGE_BUDGET = 100000          # droid_backend.py:31: effectively unlimited


class DroidBackend:
    def __init__(self, net, video, args):
        self.video = video
        self.update_op = net.update
        self.device = args.device
        self.t0 = self.t1 = 0
        # proximity-edge selection parameters (droid_backend.py:19-22)
        self.edge_rule = dict(rad=args.backend_radius, nms=args.backend_nms, thresh=args.backend_thresh, beta=args.beta)
        self.beta, self.backend_radius = args.beta, args.backend_radius
        self.backend_nms, self.backend_thresh = args.backend_nms, args.backend_thresh
        self.corr_impl = getattr(args, "backend_corr", "alt")

    def _connect_all(self):
        graph = FactorGraph(self.video, self.update_op, self.device, corr_impl=self.corr_impl, max_factors=_EDGE_BUDGET)
        graph.add_proximity_factors(**self.edge_rule)
        return graph

    @torch.no_grad()
    def __call__(self, steps=12):
        n_keyframes = self.video.counter
        self.video.normalize()
        graph = self._connect_all()
        sharded, before = None, None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one process per GPU, every rank holds the same video: each keeps the edges whose source frame it owns
            # (the selection above is deterministic, so all ranks agree on the partition) and the BA's reduced pose
            # system is all-reduced once per Gauss-Newton step (pvo_amd/parallel.py)
            from .parallel import ShardedBA, partition_by_source
            owner, _ = partition_by_source(graph._ii_h, torch.distributed.get_world_size())
            rank = torch.distributed.get_rank()
            whole = (list(graph._ii_h), list(graph._jj_h))               # the structure of the all-reduced pose system
            graph.rm_factors([o != rank for o in owner])
            sharded, before = ShardedBA(structure=whole), self.video.disps.clone()
        if len(graph._ii_h) > 65535:
            # include/pvo_hip.h "Limits": an edge index is a grid's y / z coordinate.  (The reference has no such bound;
            # a 64-keyframe window with radius 3 has 372 edges, a 1000-keyframe sequence at ~50 edges per frame would need
            # edge sharding - pvo_amd/parallel.py - or a tighter backend_thresh.)
            raise RuntimeError("global bundle adjustment over %d edges: libpvo_hip handles at most 65535 per call "
                               "(shard the edges over ranks or lower backend_thresh / backend_radius)" % len(graph._ii_h))
        if graph._ii_h:
            graph.update_lowmem(steps=steps, sharded=sharded)
        elif sharded is not None:
            raise RuntimeError("rank %d owns no edges: fewer source frames than ranks" % rank)
        if sharded is not None:
            sharded.sync_disps(self.video.disps, before)                 # every rank ends with all depth maps
        graph.clear_edges()
        self.video.dirty[:n_keyframes] = True
