"""Global bundle adjustment over every keyframe collected so far.

Role of the reference's `DroidBackend` (VO_Module/droid_slam/droid_backend.py:9-41), same constructor and call
signature.  One pass = rescale the map to unit mean inverse depth, connect all keyframes by proximity, iterate
`FactorGraph.update_lowmem`, drop the edges again.

Correlation features: `args.backend_corr = "alt"` (the reference's path: no stored volumes, features correlated on the fly
by the alt-corr HIP kernel, operator in 8-frame chunks), `"volume"`: on MI355X the volumes of every edge of the global graph
fit in HBM (25 MB per edge of 288 GB), so the global update runs like the frontend's - resident tiled pool, one native call
per step for the whole graph (~8x faster at 372 edges; values carry the volume's fp16 rounding) -, or `"auto"` (default since
round 5: "volume" when the estimate fits 60 % of the free HBM and the operator has the native path, "alt" otherwise; the
full-sequence run of bench.py spent 30 % of its kernel time in the alt-corr kernel, 1.5 ms per launch).
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
        self.corr_impl = getattr(args, "backend_corr", "auto")      # "alt" = the reference's formulation (parity runs), "auto" = the fast one that fits
        self.last_corr_impl = None                                  # what the last pass ran on ("alt" / "volume"): results record it

    def _volumes_fit(self, n_edges):
        """resident correlation volumes + per-edge operator state for n_edges, against the free HBM (288 GB on MI355X): the 4-level
        pyramid is 2.66 HW^2 bytes per edge (24 MB at 30x101), state + workspace ~5 KB per pixel"""
        hw = (self.video.ht // 8) * (self.video.wd // 8)
        need = n_edges * (2.7 * hw * hw + 5200.0 * hw)
        if torch.device(self.device).type != "cuda":
            return False
        free, _total = torch.cuda.mem_get_info(torch.device(self.device))
        free += torch.cuda.memory_reserved(torch.device(self.device)) - torch.cuda.memory_allocated(torch.device(self.device))
        return need < 0.6 * free

    def _graph(self, impl):
        return FactorGraph(self.video, self.update_op, self.device, corr_impl=impl, max_factors=_EDGE_BUDGET)

    def _connect_all(self, keep=None):
        """the global graph (droid_backend.py:31-33).  keep(ii, jj) -> mask: the edges this rank keeps (edge sharding).
        Returns (graph, whole edge lists).  With backend_corr = "auto" (default) the edges are selected on a volume-free graph
        first - selection needs no correlation features - and the graph that is optimised holds resident volumes when they
        fit in HBM and the operator has the native path, the reference's alt-corr formulation otherwise."""
        impl = self.corr_impl
        first = self._graph("alt" if impl == "auto" else impl)
        first.add_proximity_factors(**self.edge_rule)
        whole = (list(first._ii_h), list(first._jj_h))
        mask = keep(*whole) if keep is not None else None
        if impl != "auto":
            if mask is not None:
                first.rm_factors([not m for m in mask])
            return first, whole
        ii = [i for k, i in enumerate(whole[0]) if mask is None or mask[k]]
        jj = [j for k, j in enumerate(whole[1]) if mask is None or mask[k]]
        probe = self._graph("volume")
        fits = bool(ii) and probe._static_ok() and self._volumes_fit(len(ii))
        if keep is not None:
            # edge sharding: every rank judges the fit from ITS OWN free memory and edge count, but all ranks must run the same
            # formulation of the same collective solve (fp16 resident volumes and fp32 alt-corr differ numerically): the choice is
            # "volume" only if it fits on EVERY rank (MIN over ranks)
            flag = torch.tensor([1 if fits else 0], dtype=torch.int32, device=self.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            fits = bool(int(flag.item()))
        if fits:
            first.clear_edges()
            probe.add_factors(ii, jj)
            if probe._fused_ok() and probe.P_zr is not None:
                self.last_corr_impl = "volume"
                return probe, whole
            probe.clear_edges()
            first = self._graph("alt")
            first.add_factors(ii, jj)
        elif mask is not None:
            first.rm_factors([not m for m in mask])
        return first, whole

    @torch.no_grad()
    def __call__(self, steps=12):
        n_keyframes = self.video.counter
        self.video.normalize()
        sharded, before, keep = None, None, None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one process per GPU, every rank holds the same video: each keeps the edges whose source frame it owns
            # (the selection is deterministic, so all ranks agree on the partition) and the BA's reduced pose
            # system is all-reduced once per Gauss-Newton step (pvo_amd/parallel.py)
            from .parallel import ShardedBA, partition_by_source
            rank = torch.distributed.get_rank()
            keep = lambda ii, jj: [o == rank for o in partition_by_source(ii, torch.distributed.get_world_size())[0]]
        self.last_corr_impl = "alt"
        graph, whole = self._connect_all(keep)
        if self.corr_impl != "auto":
            self.last_corr_impl = self.corr_impl
        if keep is not None:
            sharded, before = ShardedBA(structure=whole), self.video.disps.clone()   # `whole`: the structure of the all-reduced pose system
        if len(graph._ii_h) > 65535:
            # include/pvo_hip.h "Limits": an edge index is a grid's y / z coordinate.  (The reference has no such bound;
            # a 64-keyframe window with radius 3 has 372 edges, a 1000-keyframe sequence at ~50 edges per frame would need
            # edge sharding - pvo_amd/parallel.py - or a tighter backend_thresh.)
            raise RuntimeError("global bundle adjustment over %d edges: libpvo_hip handles at most 65535 per call "
                               "(shard the edges over ranks or lower backend_thresh / backend_radius)" % len(graph._ii_h))
        if graph._ii_h:
            graph.update_lowmem(steps=steps, sharded=sharded)
        elif sharded is not None:
            raise RuntimeError("rank %d owns no edges: fewer source frames than ranks" % torch.distributed.get_rank())
        if sharded is not None:
            sharded.sync_disps(self.video.disps, before)                 # every rank ends with all depth maps
        graph.clear_edges()
        self.video.dirty[:n_keyframes] = True
