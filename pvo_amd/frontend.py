"""DroidFrontend — local-window keyframe optimisation (the caller of the hot path).

Counterpart of the reference's DroidFrontend (VO_Module/droid_slam/droid_frontend.py:9-112): same constants
(max_factors 48, max_age 25, iters1 4, iters2 2), same initialise / update sequence.  `video.counter` is a plain
int here (one process per GPU) and the keyframe-distance test reads ONE scalar back per keyframe — the only
host synchronisation left in a keyframe update.  The update is written in two halves around that read-back (`begin` / `finish`):
called back to back they are the reference's sequence; `Droid(args.pipelined=True)` runs the second half of keyframe t inside
track(t + 1), behind the launch of frame t + 1's encoder graph, so the device has work queued while the host waits and books.
"""
import torch

from .factor_graph import FactorGraph


class DroidFrontend:
    def __init__(self, update_op, video, device="cuda:0", warmup=8, beta=0.3, frontend_nms=1, keyframe_thresh=4.0,
                 frontend_window=25, frontend_thresh=16.0, frontend_radius=2, max_factors=48):
        self.video, self.update_op = video, update_op
        self.graph = FactorGraph(video, update_op, device, max_factors=max_factors)
        self.t0 = self.t1 = 0
        self.is_initialized = False
        self.count = 0
        self.max_age, self.iters1, self.iters2 = 25, 4, 2
        self.warmup, self.beta, self.frontend_nms = warmup, beta, frontend_nms
        self.keyframe_thresh, self.frontend_window = keyframe_thresh, frontend_window
        self.frontend_thresh, self.frontend_radius = frontend_thresh, frontend_radius
        self.keyframe_decision = None
        self.keyframes_removed = 0
        self.update_pending = False                 # a keyframe update whose second half has not run yet (pipelined Droid)
        self.prefetch = True                        # launch an update's proximity distances ahead of the keyframe's context encoder
        self._dist = self._dist_host = self._dist_ready = None

    def _update(self):
        """add edges, optimise, decide whether the previous frame stays a keyframe (droid_frontend.py:36-70)"""
        self._update_begin()
        self._update_finish()

    def _update_begin(self):
        """first half of a keyframe update (droid_frontend.py:36-52): edges in, four graph updates, the keyframe test's distance
        LAUNCHED - its scalar goes to a pinned host buffer behind an event and is read in `_update_finish`.  `Droid` in pipelined
        mode returns from track() here and runs the second half inside the next call, behind the next frame's encoder launch."""
        self.count += 1
        self.t1 += 1
        if self.graph.corr is not None:                      # droid_frontend.py:42-43
            self.graph.rm_factors([a > self.max_age for a in self.graph._age_h], store=True)
        self.graph.add_proximity_factors(self.t1 - 5, max(self.t1 - self.frontend_window, 0), rad=self.frontend_radius,
                                         nms=self.frontend_nms, thresh=self.frontend_thresh, beta=self.beta, remove=True)
        for _ in range(self.iters1):
            self.graph.update(None, None, use_inactive=True)
        d = self.video.distance([self.t1 - 3], [self.t1 - 2], beta=self.beta, bidirectional=True)
        if d.is_cuda:
            if self._dist_host is None:
                self._dist_host = torch.empty(1, dtype=d.dtype).pin_memory()
                self._dist_ready = torch.cuda.Event()
            self._dist_host.copy_(d.reshape(-1)[:1], non_blocking=True)
            self._dist_ready.record()
            self._dist = None
        else:
            self._dist = d
        self.update_pending = True

    def _update_finish(self):
        """second half (droid_frontend.py:53-70): read the distance, drop the keyframe or refine twice more, seed the next frame"""
        if not self.update_pending:
            return
        self.update_pending = False
        # (keyframe_decision: measurement / test hook - a callable (update index, distance as a float) -> True to drop the keyframe.
        # With random-init weights the distance is chaotic; bench.py and the GPU tests feed a seeded schedule so that the removal
        # branch - rm_keyframe, the counter / t1 roll-back - runs the same way in every pass.  The scalar is read back either way.)
        if self._dist is None:
            self._dist_ready.synchronize()
            dist = float(self._dist_host[0])
        else:
            dist = self._dist.item()
        drop = self.keyframe_decision(self.count, dist) if self.keyframe_decision is not None else dist < self.keyframe_thresh
        if drop:
            self.graph.rm_keyframe(self.t1 - 2)
            self.video.counter -= 1
            self.t1 -= 1
            self.keyframes_removed += 1
        else:
            for _ in range(self.iters2):
                self.graph.update(None, None, use_inactive=True)
        self.video.poses[self.t1] = self.video.poses[self.t1 - 1]
        self.video.disps[self.t1] = self.video.disps[self.t1 - 1].mean()
        self.video.dirty[min(self.graph._ii_h):self.t1] = True

    def _initialize(self):
        """bootstrap on the first `warmup` keyframes (droid_frontend.py:72-101)"""
        self.t0, self.t1 = 0, self.video.counter
        self.graph.add_neighborhood_factors(self.t0, self.t1, r=3)
        for _ in range(8):
            self.graph.update(1, use_inactive=True)
        self.graph.add_proximity_factors(0, 0, rad=2, nms=2, thresh=self.frontend_thresh)
        for _ in range(12):
            self.graph.update(1, use_inactive=True)
        self.video.poses[self.t1] = self.video.poses[self.t1 - 1].clone()
        self.video.disps[self.t1] = self.video.disps[self.t1 - 4:self.t1].mean()
        self.is_initialized = True
        self.video.dirty[:self.t1] = True

    def __call__(self):
        self.finish()
        if not self.is_initialized and self.video.counter == self.warmup:
            self._initialize()
        elif self.is_initialized and self.t1 < self.video.counter:
            self._update()

    def begin(self):
        """everything of this frame's frontend work that can be issued without the keyframe test's answer"""
        self.finish()
        if not self.is_initialized and self.video.counter == self.warmup:
            self._initialize()
        elif self.is_initialized and self.t1 < self.video.counter:
            self._update_begin()

    def finish(self):
        if self.update_pending:
            self._update_finish()

    def keyframe_ahead(self):
        """the motion filter has decided that the frame in its hands becomes a keyframe (MotionFilter.before_context): if a keyframe update
        will follow its append, launch that update's proximity distances now, in front of the context encoder (prefetch_proximity)"""
        if self.is_initialized and not self.update_pending and self.t1 == self.video.counter and self.prefetch:
            t1 = self.t1 + 1
            self.graph.prefetch_proximity(t1 - 5, max(t1 - self.frontend_window, 0), self.video.counter + 1, self.beta)
