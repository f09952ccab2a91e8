"""SE3 — the subset of lietorch the VO hot path uses, as differentiable torch ops.

Mirrors `lietorch.SE3` (reference thirdparty/lietorch/lietorch/groups.py:51-231,
265-284; math from include/se3.h:36-56,84-86,124-142 and include/so3.h:55-60,
115-208): data layout [..., 7] = (tx,ty,tz, qx,qy,qz,qw); `*` composes groups or
acts on homogeneous points [..., 4]; `inv`, `adjT`, `exp`, `log`, `retr`.

True broadcasting is used (lietorch materialises the pose once per pixel with
`.repeat`, broadcasting.py:27-29).  Two implementations of the same formulas:
  * on the GPU: ONE fused HIP kernel per operation (pvo_amd/csrc/se3_ops.hip, `pvo_se3_unary/_binary`; lietorch's
    counterpart is its CUDA element-wise kernel set, lietorch_gpu.cu:21-296), a smaller operand indexed i // rep - and,
    since round 4, ONE kernel for its backward pass (`pvo_se3_*_vjp` behind torch.autograd.Function: lietorch's backward
    kernels), where the torch formulation costs ~25 element-wise launches forward and ~50 backward per operation - 46 % of the
    operators a training step dispatches (profiles/r04_train_step_stats.txt);
  * otherwise (CPU, general broadcasting, config knob "se3_torch") the torch ops below, whose autograd is also the reference the
    backward kernels are tested against (tests/test_se3.py).
The native BA / reprojection kernels do not go through this class; it serves the
differentiable Python path (geom/ba.py, DroidNet.forward) and host-side bookkeeping.
"""
import torch

EPS = 1e-6  # lietorch include/common.h:7


FORCE_TORCH = False      # tests: the torch formulation everywhere (also pvo_amd.config "se3_torch"; nothing is read from the environment)


def _native(*ts):
    """the fused HIP kernels apply: device tensors of one fp32 / fp64 dtype"""
    t0 = ts[0]
    if FORCE_TORCH or not t0.is_cuda or t0.dtype not in (torch.float32, torch.float64):
        return False
    for t in ts:
        if not t.is_cuda or t.dtype != t0.dtype or t.device != t0.device:
            return False
    return True


def _grad(*ts):
    return torch.is_grad_enabled() and any(t.requires_grad for t in ts)


class _Unary(torch.autograd.Function):
    """exp / log / inv: forward and backward one HIP kernel each"""

    @staticmethod
    def forward(ctx, op, x):
        from .. import droid_backends as db
        x = x.contiguous()
        ctx.op = op
        ctx.save_for_backward(x)
        return db.se3_unary(op, x)

    @staticmethod
    def backward(ctx, gy):
        from .. import droid_backends as db
        (x,) = ctx.saved_tensors
        return None, db.se3_unary_vjp(ctx.op, x, gy.contiguous())


class _Binary(torch.autograd.Function):
    """mul / act / adj / adjT with index broadcasting; the gradient of a broadcast operand is summed over its repeats"""

    @staticmethod
    def forward(ctx, op, a, rep_a, b, rep_b, out_shape):
        from .. import droid_backends as db
        a, b = a.contiguous(), b.contiguous()
        ctx.op, ctx.rep_a, ctx.rep_b = op, rep_a, rep_b
        ctx.save_for_backward(a, b)
        return db.se3_binary(op, a, rep_a, b, rep_b, out_shape)

    @staticmethod
    def backward(ctx, gy):
        from .. import droid_backends as db
        a, b = ctx.saved_tensors
        ga, gb = db.se3_binary_vjp(ctx.op, a, ctx.rep_a, b, ctx.rep_b, gy.contiguous(), ctx.needs_input_grad[1], ctx.needs_input_grad[3])
        return None, ga, None, gb, None, None


def _unary(op, x):
    from .. import droid_backends as db
    return _Unary.apply(op, x) if _grad(x) else db.se3_unary(op, x.contiguous())


def _bcast(sa, sb):
    """batch shapes sa, sb -> (out shape, rep_a, rep_b) when the broadcast is an index broadcast (one operand's batch is the
    other's with trailing 1s, or the two are equal); None otherwise"""
    n = max(len(sa), len(sb))
    sa = (1,) * (n - len(sa)) + tuple(sa)
    sb = (1,) * (n - len(sb)) + tuple(sb)
    if sa == sb:
        return sa, 1, 1
    for x, y, swap in ((sa, sb, False), (sb, sa, True)):
        k = n
        while k > 0 and x[k - 1] == 1:
            k -= 1
        if x[:k] == y[:k]:
            rep = 1
            for d in y[k:]:
                rep *= d
            return (y, 1, rep) if swap else (y, rep, 1)
    return None


def _binary(op, g, b):
    """native group-on-vector / group-on-group operation, or None when the shapes need general broadcasting"""
    from .. import droid_backends as db
    bc = _bcast(g.shape[:-1], b.shape[:-1])
    if bc is None:
        return None
    out, rep_a, rep_b = bc
    if _grad(g, b):
        return _Binary.apply(op, g, rep_a, b, rep_b, out)
    return db.se3_binary(op, g.contiguous(), rep_a, b.contiguous(), rep_b, out)


def _cross(a, b):
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], dim=-1)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def _qconj(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def _qrot(q, v):
    qv = q[..., :3]
    uv = 2.0 * _cross(qv, v)
    return v + q[..., 3:] * uv + _cross(qv, uv)


def _so3_exp(phi):
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = torch.sqrt(th2)
    small = th < EPS
    th_safe = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th2 / 48.0 + th2 * th2 / 3840.0, torch.sin(0.5 * th_safe) / th_safe)
    real = torch.where(small, 1.0 - th2 / 8.0 + th2 * th2 / 384.0, torch.cos(0.5 * th_safe))
    return torch.cat([imag * phi, real], dim=-1)


def _so3_log(q):
    v, w = q[..., :3], q[..., 3:]
    n2 = (v * v).sum(-1, keepdim=True)
    small = n2 < EPS * EPS
    n = torch.sqrt(torch.where(small, torch.ones_like(n2), n2))
    w_safe = torch.where(w.abs() < EPS, torch.full_like(w, EPS), w)
    big = 2.0 * torch.atan(n / w_safe) / n
    near_pi = torch.where(w > 0, 3.14159265358979323846 / n, -3.14159265358979323846 / n)
    big = torch.where(w.abs() < EPS, near_pi, big)
    sm = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w)
    return torch.where(small, sm, big) * v


def _left_jacobian_coefs(phi):
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = torch.sqrt(th2)
    small = th < EPS
    th2s = torch.where(small, torch.ones_like(th2), th2)
    ths = torch.where(small, torch.ones_like(th), th)
    c1 = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / th2s)
    c2 = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (th2s * ths))
    return c1, c2


class SE3:
    group_name = "SE3"
    group_id = 3
    manifold_dim = 6
    embedded_dim = 7
    id_elem = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]

    def __init__(self, data):
        if isinstance(data, SE3):
            data = data.data
        self.data = data

    # ---- lietorch.LieGroup surface ------------------------------------------------
    def __repr__(self):
        return "SE3: size=%s, device=%s, dtype=%s" % (tuple(self.shape), self.device, self.dtype)

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    def vec(self):
        return self.data

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (6,)

    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if len(batch_shape) == 1 and isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        data = torch.as_tensor(cls.id_elem, **kwargs)
        data = data.view((1,) * len(batch_shape) + (7,)).expand(tuple(batch_shape) + (7,)).contiguous()
        return cls(data)

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def InitFromVec(cls, data):
        return cls(data)

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        if len(batch_shape) == 1 and isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        return cls.exp(sigma * torch.randn(tuple(batch_shape) + (6,), **kwargs))

    @classmethod
    def exp(cls, x):
        if _native(x):
            return cls(_unary("exp", x))
        tau, phi = x[..., :3], x[..., 3:]
        q = _so3_exp(phi)
        c1, c2 = _left_jacobian_coefs(phi)
        pt = _cross(phi, tau)
        t = tau + c1 * pt + c2 * _cross(phi, pt)
        return cls(torch.cat([t, q], dim=-1))

    def log(self):
        if _native(self.data):
            return _unary("log", self.data)
        t, q = self.data[..., :3], self.data[..., 3:]
        phi = _so3_log(q)
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = torch.sqrt(th2)
        small = th < EPS
        ths = torch.where(small, torch.ones_like(th), th)
        half = 0.5 * ths
        c2 = torch.where(small, torch.full_like(th, 1.0 / 12.0),
                         (1.0 - ths * torch.cos(half) / (2.0 * torch.sin(half))) / (ths * ths))
        pt = _cross(phi, t)
        tau = t - 0.5 * pt + c2 * _cross(phi, pt)
        return torch.cat([tau, phi], dim=-1)

    def inv(self):
        if _native(self.data):
            return SE3(_unary("inv", self.data))
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = _qconj(q)
        return SE3(torch.cat([-_qrot(qi, t), qi], dim=-1))

    def mul(self, other):
        if _native(self.data, other.data):
            y = _binary("mul", self.data, other.data)
            if y is not None:
                return SE3(y)
        t1, q1 = self.data[..., :3], self.data[..., 3:]
        t2, q2 = other.data[..., :3], other.data[..., 3:]
        return SE3(torch.cat([t1 + _qrot(q1, t2), _qmul(q1, q2)], dim=-1))

    def retr(self, a):
        return SE3.exp(a).mul(self)

    def act(self, p):
        if p.shape[-1] in (3, 4) and _native(self.data, p):
            y = _binary("act4" if p.shape[-1] == 4 else "act3", self.data, p)
            if y is not None:
                return y
        t, q = self.data[..., :3], self.data[..., 3:]
        if p.shape[-1] == 3:
            return _qrot(q, p) + t
        w = p[..., 3:]
        xyz = _qrot(q, p[..., :3]) + t * w
        return torch.cat([xyz, w.expand(xyz.shape[:-1] + (1,))], dim=-1)

    def adj(self, a):
        if _native(self.data, a):
            y = _binary("adj", self.data, a)
            if y is not None:
                return y
        t, q = self.data[..., :3], self.data[..., 3:]
        Rphi = _qrot(q, a[..., 3:])
        return torch.cat([_qrot(q, a[..., :3]) + _cross(t, Rphi), Rphi], dim=-1)

    def adjT(self, a):
        if _native(self.data, a):
            y = _binary("adjT", self.data, a)
            if y is not None:
                return y
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = _qconj(q)
        a_tau, a_phi = a[..., :3], a[..., 3:]
        return torch.cat([_qrot(qi, a_tau), _qrot(qi, a_phi) + _qrot(qi, _cross(a_tau, t))], dim=-1)

    def matrix(self):
        I = torch.eye(4, dtype=self.dtype, device=self.device)
        I = I.view([1] * (len(self.data.shape) - 1) + [4, 4])
        return SE3(self.data[..., None, :]).act(I).transpose(-1, -2)

    def translation(self):
        p = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=self.dtype, device=self.device)
        return self.act(p.view([1] * (len(self.data.shape) - 1) + [4]))

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        return SE3(torch.cat([t * s.unsqueeze(-1), q], dim=-1))

    def detach(self):
        return SE3(self.data.detach())

    def view(self, dims):
        return SE3(self.data.view(tuple(dims) + (7,)))

    def __mul__(self, other):
        if isinstance(other, SE3):
            return self.mul(other)
        return self.act(other)

    def __getitem__(self, index):
        return SE3(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def to(self, *args, **kwargs):
        return SE3(self.data.to(*args, **kwargs))

    def cpu(self):
        return SE3(self.data.cpu())

    def cuda(self):
        return SE3(self.data.cuda())

    def float(self, device=None):
        return SE3(self.data.float())

    def double(self, device=None):
        return SE3(self.data.double())

    def unbind(self, dim=0):
        return [SE3(x) for x in self.data.unbind(dim=dim)]


def cat(group_list, dim):
    return SE3(torch.cat([g.data for g in group_list], dim=dim))


def stack(group_list, dim):
    return SE3(torch.stack([g.data for g in group_list], dim=dim))
