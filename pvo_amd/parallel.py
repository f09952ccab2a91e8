"""Edge-sharded bundle adjustment across GPUs.

Factor-graph edges are partitioned BY SOURCE KEYFRAME: all edges i->* live on the owner of i.
Then everything that couples edges through a depth map — C, w, Q, the Ei rows, every Schur term
(both edges of an (a,b,k) triple share the source frame k, droid_kernels.cu:1236-1246) and the
depth update itself — is rank-local, reprojection i->j only needs disps[i], and depth maps never
move.  Poses (7 floats each) are replicated.  Per Gauss-Newton step there is exactly ONE
collective: all-reduce (sum) of the reduced pose system [(6P)^2 + 6P] fp64 between
`pvo_ba_local` and `pvo_ba_finish`; every rank then factorises the identical system, so the pose
replicas stay bit-identical without a broadcast.  The HIP library keeps that system in 64-bit FIXED POINT, so the
all-reduce is an integer sum: exact, order independent, and equal to what one GPU computes on the whole graph.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI: the message is
18.8 KB at P=8, 90 KB at P=25 — pure latency, one all-reduce per step, not per edge).
The reference has no multi-GPU inference path; this is new design, not a translation.
"""
import torch
import torch.distributed as dist


def partition_by_source(ii, world_size):
    """owner rank of every edge: greedy longest-processing-time assignment of source frames by
    their out-degree (ties broken by frame id, so every rank computes the same partition)."""
    ii = [int(v) for v in ii]
    deg = {}
    for f in ii:
        deg[f] = deg.get(f, 0) + 1
    load = [0] * world_size
    owner = {}
    for f in sorted(deg, key=lambda f: (-deg[f], f)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        owner[f] = r
        load[r] += deg[f]
    return [owner[f] for f in ii], owner


def local_eta_rows(ii_all, ii_local, t0, t1):
    """eta carries one row per optimised depth map, in the order of unique([t0,t1) U ii).  A rank that
    only holds some edges optimises fewer maps: indices of its rows inside the global eta."""
    kx_global = sorted(set(range(t0, t1)) | set(int(v) for v in ii_all))
    kx_local = sorted(set(range(t0, t1)) | set(int(v) for v in ii_local))
    pos = {f: k for k, f in enumerate(kx_global)}
    return [pos[f] for f in kx_local]


def envelope_structure(ii_all, jj_all, t0, t1):
    """first[b] for every free pose b in [t0, t1): the lowest free pose its block row of the reduced pose system can couple
    with - STRUCTURALLY, from the global edge list, so that every rank derives the same table without communication.
    Two poses couple through an edge between them (the Hii/Hij/Hjj blocks of droid_kernels.cu:309-357) or through a depth
    map both observe: source k and every target of an edge leaving k (the Schur terms, :1236-1246)."""
    P = t1 - t0
    first = list(range(P))
    groups = {}
    for i, j in zip(ii_all, jj_all):
        groups.setdefault(int(i), {int(i)}).add(int(j))
    for members in groups.values():
        ps = sorted(m - t0 for m in members if t0 <= m < t1)
        for p in ps[1:]:
            first[p] = min(first[p], ps[0])
    return first


def envelope_index(first, device):
    """flat positions inside `sys` = [(6P)^2 row-major | 6P rhs] of the entries that are all-reduced: the lower-triangle
    blocks (b, first[b] .. b) and the right-hand side.  Cholesky reads the lower triangle only and creates no fill outside
    the envelope, and every entry outside it is a structural zero on EVERY rank - so reducing just these entries gives the
    same factorisation, bit for bit, as reducing the dense system (63 free poses of a radius-3 graph: 127 KB instead of
    1.15 MB)."""
    P = len(first)
    n6 = 6 * P
    if P == 0:
        return torch.zeros(0, dtype=torch.long, device=device)
    # (tensor arithmetic, not Python loops: a global BA over several hundred free poses has tens of millions of entries)
    fb = torch.as_tensor(list(first), dtype=torch.long)
    b = torch.arange(P, dtype=torch.long)
    ncols = 6 * (b - fb + 1)                                   # scalar columns of block row b inside the envelope
    rows = torch.arange(n6, dtype=torch.long)                  # every scalar row 6 b + r ...
    cnt = ncols.repeat_interleave(6)                           # ... holds ncols[b] consecutive entries starting at column 6 first[b]
    start = rows * n6 + (6 * fb).repeat_interleave(6)
    offs = torch.arange(int(cnt.sum()), dtype=torch.long) - (torch.cumsum(cnt, 0) - cnt).repeat_interleave(cnt)
    idx = torch.cat([start.repeat_interleave(cnt) + offs, torch.arange(n6 * n6, n6 * n6 + n6, dtype=torch.long)])
    return idx.to(device)


class ShardedBA:
    """Dense BA over this rank's edge shard.  `backend` supplies the three native steps
    (default: pvo_amd.droid_backends); the oracle-backed variant in tests/ exercises the same
    partition / reduction logic on CPU with gloo."""

    def __init__(self, group=None, backend=None, structure=None, communicate=True):
        self.group = group
        self.communicate = communicate  # False: this process holds the whole graph although a process group exists (bench reference)
        self.structure = structure      # (ii_all, jj_all) of the whole graph: default of ba()'s `structure`
        if backend is None:
            from . import droid_backends as backend
        self.db = backend
        self._ws = None
        self._plan_key = None
        self._env_idx = None
        self._sys_clean = False
        self.always_pack = False        # tests: run the pack / unpack pair on a single rank too
        self.torch_pack = False         # tests: index_select / index_copy_ around the collective although the backend packs natively
        self.collective_at_one = False  # run the all-reduce through the process group even when it has ONE rank (the RCCL path on a 1-GPU box)
        self.last_message_bytes = 0

    # the two places that touch the process group (tests substitute an in-process exchange)
    def _world(self):
        if not (self.communicate and dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def _allreduce(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def ba(self, poses, disps, intrinsics, targets, weights, eta_local, ii_local, jj_local, t0, t1,
           itrs=2, lm=1e-4, ep=0.1, motion_only=False, plan_key=None, structure=None):
        """poses/disps updated in place (disps: only the maps this rank owns change).
        targets/weights/ii/jj/eta_local describe THIS RANK's edges (see partition_by_source,
        local_eta_rows).  Returns dx [P,6] of the last step.
        plan_key: any hashable that identifies the edge set (e.g. (id(graph), graph._version)); while it stays the same
        the BA plan and the system buffer of the previous call are reused instead of rebuilt.
        structure: (ii_all, jj_all) host lists of the WHOLE graph's edges (every rank knows them: the partition is computed
        from them).  When given, only the envelope of the pose system is all-reduced (`envelope_index`) instead of the
        dense (6P)^2 matrix; the result is bit-identical."""
        F, ht, wd = disps.shape
        P = t1 - t0
        E = ii_local.shape[0]
        n6 = 6 * P
        if structure is None:
            structure = self.structure
        if self._ws is None or self._ws[0] != (E, P, F, ht * wd):
            self._ws = ((E, P, F, ht * wd), self.db.ba_workspace(E, P, F, ht * wd, disps.device))
        ws = self._ws[1]
        K_eta = -1 if motion_only else eta_local.reshape(-1, ht * wd).shape[0]
        key = (E, P, F, ht * wd, K_eta, t0, t1, plan_key)
        if plan_key is None or self._plan_key != key:      # the plan and the system buffer belong to the edge set, not to the call
            self._sys = torch.zeros(n6 * n6 + n6, dtype=getattr(self.db, "BA_SYS_DTYPE", torch.float64), device=disps.device)
            self._sys_clean = True                      # (the HIP library's finish / pack leave the buffer zeroed: no memset per call either)
            self.db.ba_plan(ii_local, jj_local, F, ht * wd, K_eta, t0, t1, ws)
            self._plan_key, self._plan_edges = key, (ii_local, jj_local)
            self._env_idx = None
        multi = self._world() > 1 or (self.collective_at_one and self.communicate and dist.is_available() and dist.is_initialized())
        native = hasattr(self.db, "ba_pack") and not self.torch_pack      # the HIP library packs / unpacks the message itself
        if getattr(self, "_env_native", native) != native:                # (torch_pack / db changed between two calls on one plan:
            self._env_idx = None                                          # the cached message index has the other form)
        self._env_native = native
        if structure is not None and self._env_idx is None and (multi or self.always_pack):
            first = envelope_structure(structure[0], structure[1], t0, t1)
            if native:
                self._env_idx = (torch.tensor(first, dtype=torch.int32, device=disps.device),
                                 torch.empty(self.db.ba_packed_elems(first), dtype=torch.int64, device=disps.device))
            else:
                self._env_idx = envelope_index(first, disps.device)
        sys_buf = self._sys
        dx = None
        ii_local, jj_local = self._plan_edges          # the tensors the plan was built from
        fixed = hasattr(self.db, "BA_SYS_DTYPE")       # the HIP library: finish leaves the buffer zeroed
        for it in range(itrs):
            kw = {"sys_is_zero": True} if fixed and self._sys_clean else {}
            self._sys_clean = False                     # (until this step's finish has run)
            self.db.ba_local(poses, disps, intrinsics, targets, weights, eta_local, ii_local, jj_local, t0, t1,
                             motion_only, sys_buf, ws, **kw)
            if structure is not None and self._env_idx is not None and native:
                first_dev, msg = self._env_idx
                self.db.ba_pack(sys_buf, first_dev, msg)                           # envelope blocks + rhs -> msg, sys_buf zeroed: one launch
                if multi:
                    self._allreduce(msg)                                           # the one collective per step
                self.last_message_bytes = msg.numel() * msg.element_size()
                dx, _ = self.db.ba_finish(poses, disps, sys_buf, ii_local, jj_local, t0, t1, lm, ep, motion_only, ws, packed=(msg, first_dev))
                self._sys_clean = fixed
                continue
            if structure is not None and self._env_idx is not None:
                msg = sys_buf.index_select(0, self._env_idx)                       # envelope blocks + rhs: one gather
                if multi:
                    self._allreduce(msg)                                           # the one collective per step
                sys_buf.index_copy_(0, self._env_idx, msg)
                self.last_message_bytes = msg.numel() * msg.element_size()
            elif multi:
                self._allreduce(sys_buf)                                           # the one collective per step
                self.last_message_bytes = sys_buf.numel() * sys_buf.element_size()
            dx, _ = self.db.ba_finish(poses, disps, sys_buf, ii_local, jj_local, t0, t1, lm, ep, motion_only, ws)
            self._sys_clean = fixed
        return dx

    def sync_disps(self, disps, disps_before):
        """make the depth replicas identical again (only needed when something reads maps a rank does
        not own, e.g. writing results): all-reduce of the per-rank updates."""
        delta = disps - disps_before
        if self._world() > 1:
            self._allreduce(delta)
        disps.copy_(disps_before + delta)
        return disps


def shard_edges(ii, jj, world_size, rank):
    """this rank's share of a factor graph under the source-frame partition: (ii_local, jj_local, edge indices)"""
    owner, _ = partition_by_source(ii, world_size)
    keep = [k for k, o in enumerate(owner) if o == rank]
    ii, jj = [int(v) for v in ii], [int(v) for v in jj]
    return [ii[k] for k in keep], [jj[k] for k in keep], keep
