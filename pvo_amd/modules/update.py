"""Update operator of the VO hot path: ConvGRU + heads + graph aggregation.

State-dict compatible with the reference's `DynamicUpdateModule`
(VO_Module/droid_slam/droid_net.py:166-314), `ConvGRU` (modules/gru.py:5-34), `GraphAgg`
(droid_net.py:64-95) and `GradientClip` (modules/clipping.py:7-23): same parameter names,
shapes and creation order, so a reference checkpoint's `update.*` tensors load unchanged
and a seeded default init reproduces the reference's weights.

What differs is how the forward pass is issued on MI355X (16-bit inference path, `_forward_fused`):
the whole operator is ONE call into libpvo_hip (`pvo_update_operator`, pvo_amd/csrc/update_exec.hip) that enqueues ~17
hand-written kernels - no MIOpen, no hipBLASLt, no element-wise PyTorch launches:
  * correlation lookup + corr_encoder[0] fused (or corr_encoder[0] on a sampled tensor), flow_encoder[0] as a 7x7
    matrix-core kernel, the encoders' second layers writing straight into the ConvGRU's input buffer;
  * static-input split: conv(W,[net|inp|corr|flow]) = conv(W_dyn,[net|corr|flow]) + conv(W_inp, inp); `inp` is constant
    over an edge's life, so the second term is computed once per edge (`static_terms`) and added in the gate epilogues;
  * z and r gates as one 256-output convolution, the gate arithmetic and the state update as the convolutions' epilogues;
  * the four heads' first stages as one 512-output convolution, their second stages as one kernel;
  * GraphAgg (conv1, mean per source frame, conv2, eta head, upsampling mask) on a second stream beside the heads.
Filters are re-arranged once (`packed_weights`).  Training / fp32 / CPU keep the reference's per-layer formulation.
The reference's forward() also evaluates `np.range(...)` at droid_net.py:295, which does not
exist in NumPy (AttributeError); that dead statement is not reproduced.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

GRAD_CLIP = 0.01


class _ClipGrad(torch.autograd.Function):
    """identity forward; backward zeroes gradients with |g| > GRAD_CLIP or NaN (clipping.py:7-18)"""

    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, g):
        zero = torch.zeros_like(g)
        g = torch.where(g.abs() > GRAD_CLIP, zero, g)
        return torch.where(torch.isnan(g), zero, g)


class GradientClip(nn.Module):
    def forward(self, x):
        return _ClipGrad.apply(x)


def scatter_mean(src, index, dim, dim_size=None):
    """torch_scatter.scatter_mean (droid_net.py:87) with index_add_."""
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    out = torch.zeros(shape, dtype=src.dtype, device=src.device).index_add_(dim, index, src)
    cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add_(
        0, index, torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
    view = [1] * src.dim()
    view[dim] = dim_size
    return out / cnt.clamp(min=1).view(view)


from .. import config as _config

# A/B switches of the fused path (pvo_amd.config, set by code; part of packed_weights()' cache key):
#   conv128_wide      corr_encoder[2] / agg.conv1 on the wide-layer kernel
#   agg_side_stream   the aggregation branch (conv1 over the edges, mean per source frame, then small kernels over the K keyframes that
#                     leave most of the chip idle) on a second HIP stream beside the heads, which only share its input
#   enc_side_stream   the flow encoder, the global-context reduction and the gate context do not depend on the correlation features:
#                     second stream beside the HBM-bound lookup and corr_encoder[2]


class PoolLookup:
    """correlation features for the fused operator: a resident tiled volume pool + the coordinates to sample it at
    (`CorrVolumePool.at(coords)`); the lookup then runs fused with corr_encoder[0] inside libpvo_hip"""

    def __init__(self, levels, slots, num_slots, coords):
        self.levels, self.slots, self.num_slots, self.coords = levels, slots, num_slots, coords


class ConvGRU(nn.Module):
    def __init__(self, h_planes=128, i_planes=128):
        super().__init__()
        self.do_checkpoint = False
        self.convz = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.convr = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.convq = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.w = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self.convz_glo = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self.convr_glo = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self.convq_glo = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self._fused = None

    def train(self, mode=True):
        self._fused = None
        return super().train(mode)

    def _fused_zr(self):
        if self._fused is None or self._fused[0].dtype != self.convz.weight.dtype \
                or self._fused[0].device != self.convz.weight.device:
            w = torch.cat([self.convz.weight, self.convr.weight], 0).contiguous(memory_format=torch.channels_last)
            b = torch.cat([self.convz.bias, self.convr.bias], 0)
            wg = torch.cat([self.convz_glo.weight, self.convr_glo.weight, self.convq_glo.weight], 0)
            bg = torch.cat([self.convz_glo.bias, self.convr_glo.bias, self.convq_glo.bias], 0)
            self._fused = (w.detach(), b.detach(), wg.detach(), bg.detach())
        return self._fused

    def forward(self, net, *inputs):
        inp = torch.cat(inputs, dim=1)
        net_inp = torch.cat([net, inp], dim=1)
        b, c, h, w = net.shape
        # global context: spatial mean of sigmoid(w(net)) * net   (gru.py:22-24)
        glo = (torch.sigmoid(self.w(net)) * net).view(b, c, h * w).mean(-1).view(b, c, 1, 1)
        if self.training or torch.is_grad_enabled():
            z = torch.sigmoid(self.convz(net_inp) + self.convz_glo(glo))
            r = torch.sigmoid(self.convr(net_inp) + self.convr_glo(glo))
            q = torch.tanh(self.convq(torch.cat([r * net, inp], dim=1)) + self.convq_glo(glo))
        else:
            wz, bz, wg, bg = self._fused_zr()
            g = F.conv2d(glo, wg, bg)                       # the three 1x1 context convs at once
            zr = torch.sigmoid(F.conv2d(net_inp, wz, bz, padding=1) + g[:, :2 * c])
            z, r = zr[:, :c], zr[:, c:]
            q = torch.tanh(self.convq(torch.cat([r * net, inp], dim=1)) + g[:, 2 * c:])
        return (1 - z) * net + z * q


class GraphAgg(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(128, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 128, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.eta = nn.Sequential(nn.Conv2d(128, 1, 3, padding=1), GradientClip(), nn.Softplus())
        self.upmask_disp = nn.Sequential(nn.Conv2d(128, 8 * 8 * 9, 1, padding=0))

    def forward(self, net, ii):
        batch, num, ch, ht, wd = net.shape
        net = net.reshape(batch * num, ch, ht, wd)
        _, ix = torch.unique(ii, return_inverse=True)
        net = self.relu(self.conv1(net)).view(batch, num, 128, ht, wd)
        net = scatter_mean(net, ix, dim=1).view(-1, 128, ht, wd)   # mean over edges sharing a source frame
        net = self.relu(self.conv2(net))
        # softplus in fp32, as autocast does for the reference (softplus is on its fp32 list)
        eta = self.eta[2](self.eta[1](self.eta[0](net).float())).view(batch, -1, ht, wd)
        upmask = self.upmask_disp(net).view(batch, -1, 8 * 8 * 9, ht, wd)
        return 0.01 * eta, upmask, None, None                      # droid_net.py:95


def _head(cout):
    return nn.Sequential(nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True),
                         nn.Conv2d(128, cout, 3, padding=1), GradientClip())


class DynamicUpdateModule(nn.Module):
    def __init__(self, use_aff_bri=False):
        super().__init__()
        cor_planes = 4 * (2 * 3 + 1) ** 2
        self.mask_num = 2
        # creation order follows droid_net.py:172-225 so that a seeded init matches the reference
        self.corr_encoder = nn.Sequential(nn.Conv2d(cor_planes, 128, 1, padding=0), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True))
        self.flow_encoder = nn.Sequential(nn.Conv2d(4 + self.mask_num + 2, 128, 7, padding=3), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, 64, 3, padding=1), nn.ReLU(inplace=True))
        self.weight = _head(2)
        self.delta = _head(2)
        self.delta_dy = _head(2)
        self.delta_mask = _head(self.mask_num)
        if use_aff_bri:
            self.global_avg_pool = nn.Sequential(nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True),
                                                 nn.AdaptiveAvgPool2d((1, 1)), GradientClip())
            self.param_linear = nn.Sequential(nn.Linear(128, 2), nn.Sigmoid())
        self.use_aff_bri = use_aff_bri
        self.gru = ConvGRU(128, 128 + 128 + 64)
        self.agg = GraphAgg()
        self._fused_heads = None
        self.fused_gru = True          # 16-bit inference runs as one call into libpvo_hip (pvo_update_operator)
        # re-arranged inference weights are dropped whenever weights are (re)loaded or the mode changes
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_derived())

    def _drop_derived(self):
        self.__dict__.pop("_packed", None)
        self.__dict__.pop("_param_list", None)
        self._fused_heads = None
        self.gru._fused = None

    def train(self, mode=True):
        self._drop_derived()
        return super().train(mode)

    def _apply(self, fn, *a, **kw):      # .to() / .half() / .cuda(): the packed copies follow the parameters
        self._drop_derived()
        return super()._apply(fn, *a, **kw)

    def packed_weights(self, dt):
        """the operator's parameters re-arranged once for libpvo_hip (include/pvo_hip.h pvo_update_weights): tap-major
        16-bit filters, fp32 biases, the z|r gate filters fused and split into their dynamic ([net|corr|flow]) and static
        (`inp`) input channels, the four heads' first stages side by side.  Cached; rebuilt when a parameter changes."""
        from .. import droid_backends as db
        # (the parameter OBJECTS are listed once - walking the module tree costs 0.7 ms, and this runs once per graph update;
        # .to() / .half() / train() / load_state_dict() drop the list, in-place changes show in the version counters)
        ps = self.__dict__.get("_param_list")
        if ps is None:
            ps = self.__dict__["_param_list"] = list(self.parameters())
        _CONV128_WIDE, _AGG_SIDE_STREAM, _ENC_SIDE_STREAM = (_config.get(k) for k in ("conv128_wide", "agg_side_stream", "enc_side_stream"))
        key = (dt, ps[0].device, tuple(p._version for p in ps), _CONV128_WIDE, _AGG_SIDE_STREAM, _ENC_SIDE_STREAM)
        hit = self.__dict__.get("_packed")
        if hit is not None and hit[0] == key and hit[1] is not None:
            return hit[1]
        f32 = lambda t: t.detach().float().contiguous()
        taps = lambda w: db.conv3x3_weights(w, dt)             # wide-layer kernel (fragment order)
        taps128 = lambda w: db.conv3x3_c128_weights(w, dt)     # 128-input kernel (tap-major)
        e128 = taps if _CONV128_WIDE else taps128              # corr_encoder[2] / agg.conv1 run on either
        g = self.gru
        wzr = torch.cat([g.convz.weight, g.convr.weight], 0).detach()
        wq = g.convq.weight.detach()
        dyn = lambda w: torch.cat([w[:, :128], w[:, 256:]], 1)         # input channels [net | corr | flow]
        sta = lambda w: w[:, 128:256]                                   # input channels of `inp`
        hs = (self.delta, self.delta_dy, self.weight, self.delta_mask)
        wg = torch.cat([g.convz_glo.weight, g.convr_glo.weight, g.convq_glo.weight], 0).detach().reshape(384, 128)
        bg = torch.cat([g.convz_glo.bias, g.convr_glo.bias, g.convq_glo.bias], 0).detach().float()
        bconv = torch.cat([g.convz.bias, g.convr.bias, g.convq.bias], 0).detach().float()
        t = {
            "enc0_w": db.corr_encoder_weights(self.corr_encoder[0].weight, dt), "enc0_b": f32(self.corr_encoder[0].bias),
            "cenc2_w": e128(self.corr_encoder[2].weight), "cenc2_b": f32(self.corr_encoder[2].bias),
            "fenc0_w": db.conv7x7_c8_weights(self.flow_encoder[0].weight, dt), "fenc0_b": f32(self.flow_encoder[0].bias),
            "fenc2_w": taps128(self.flow_encoder[2].weight), "fenc2_b": f32(self.flow_encoder[2].bias),
            "glo_w": g.w.weight.detach().reshape(128, 128).to(dt).contiguous(), "glo_b": f32(g.w.bias),
            # g = Wg glo + bg + [bz | br | bq]: the z/r/q convolution biases are per-channel constants too
            "gate_wt": wg.float().t().contiguous(), "gate_b": (bg + bconv).contiguous(),
            "zr_w": taps(dyn(wzr)), "q_w": taps(dyn(wq)), "zr_inp_w": taps(sta(wzr)), "q_inp_w": taps(sta(wq)),
            "heads1_w": taps(torch.cat([h[0].weight for h in hs], 0).detach()),
            "heads1_b": torch.cat([f32(h[0].bias) for h in hs]).contiguous(),
            # [head][out][tap = ky*3+kx][channel] -> matrix-core fragments of the fused second stage
            "heads2_w": db.heads2_fragments(torch.stack([h[2].weight.detach().permute(0, 2, 3, 1).reshape(2, 9, 128) for h in hs]), dt),
            "heads2_b": torch.cat([f32(h[2].bias) for h in hs]).contiguous(),
            "agg1_w": e128(self.agg.conv1.weight), "agg1_b": f32(self.agg.conv1.bias),
            "agg2_w": taps128(self.agg.conv2.weight), "agg2_b": f32(self.agg.conv2.bias),
            "eta_w": self.agg.eta[0].weight.detach().permute(0, 2, 3, 1).reshape(9, 128).to(dt).contiguous(),
            "eta_b": f32(self.agg.eta[0].bias),
            "up_w": self.agg.upmask_disp[0].weight.detach().reshape(576, 128).to(dt).contiguous(),
            "up_b": f32(self.agg.upmask_disp[0].bias),
        }
        from .._lib import PVO_OP_CONV128_WIDE, PVO_OP_ENC_SIDE_STREAM, PVO_OP_SINGLE_STREAM
        flags = (PVO_OP_CONV128_WIDE if _CONV128_WIDE else 0) | (0 if _AGG_SIDE_STREAM else PVO_OP_SINGLE_STREAM) | \
            (PVO_OP_ENC_SIDE_STREAM if _ENC_SIDE_STREAM else 0)
        pw = db.PackedWeights(dt, t, flags)
        self.__dict__["_packed"] = (key, pw)
        return pw

    def static_terms(self, inp, dt=None):
        """the part of the ConvGRU's gate / candidate convolutions that only sees `inp` (constant over an edge's life):
        (P_zr [E,256,H,W], P_q [E,128,H,W]) channels-last; a factor graph computes them once per edge"""
        from .. import droid_backends as db
        dt = dt or inp.dtype
        pw = self.packed_weights(dt)
        x = inp.to(dt).contiguous(memory_format=torch.channels_last)
        return db.conv3x3(x, pw.tensors["zr_inp_w"]), db.conv3x3(x, pw.tensors["q_inp_w"])

    def _heads(self, net):
        """delta, delta_dy, weight, delta_mask, each [B,2,H,W] (PyTorch path: training, fp32, CPU)"""
        if self.training or torch.is_grad_enabled():
            return self.delta(net), self.delta_dy(net), self.weight(net), self.delta_mask(net)
        hs = (self.delta, self.delta_dy, self.weight, self.delta_mask)
        f = self._fused_heads
        if f is None or f[0].dtype != hs[0][0].weight.dtype or f[0].device != hs[0][0].weight.device:
            w1 = torch.cat([h[0].weight for h in hs], 0).contiguous(memory_format=torch.channels_last)
            b1 = torch.cat([h[0].bias for h in hs], 0)
            # second stage as ONE dense conv with a block-diagonal weight (4 x [2,128,3,3] on the diagonal of [8,512,3,3])
            w2 = torch.zeros(8, 512, 3, 3, dtype=hs[0][2].weight.dtype, device=hs[0][2].weight.device)
            for k, h in enumerate(hs):
                w2[2 * k:2 * k + 2, 128 * k:128 * k + 128] = h[2].weight.detach()
            w2 = w2.contiguous(memory_format=torch.channels_last)
            b2 = torch.cat([h[2].bias for h in hs], 0)
            f = self._fused_heads = (w1.detach(), b1.detach(), w2.detach(), b2.detach())
        x = F.relu(F.conv2d(net, f[0], f[1], padding=1), inplace=True)      # 128 -> 4*128
        y = F.conv2d(x, f[2].to(x.dtype), f[3].to(x.dtype), padding=1)      # 4 x (128 -> 2), block diagonal
        return y[:, 0:2], y[:, 2:4], y[:, 4:6], y[:, 6:8]

    @staticmethod
    def _segments_from(ii):
        """CSR of the edges grouped by source frame, groups in sorted(unique(ii)) order (one device read-back; factor
        graphs pass their host-built `agg_segments` instead)"""
        from .. import droid_backends as db
        ii_l = [int(v) for v in ii.tolist()]
        frames = sorted(set(ii_l))
        pos = {f: k for k, f in enumerate(frames)}
        buckets = [[] for _ in frames]
        for e, i in enumerate(ii_l):
            buckets[pos[i]].append(e)
        ptr, idx = [0], []
        for b in buckets:
            idx += b
            ptr.append(len(idx))
        both = db.to_device_async(ptr + idx, torch.int32, ii.device)
        return both[:len(ptr)], both[len(ptr):], len(frames)

    def _forward_fused(self, net, inp, corr, flow, ii, agg_segments, static_terms, dt, out_dim, single_stream=False):
        """16-bit inference: the whole operator is ONE call into libpvo_hip (pvo_update_operator, update_exec.hip)"""
        from .. import droid_backends as db
        E, _, ht, wd = net.shape
        cl = lambda t: t.to(dt).contiguous(memory_format=torch.channels_last)
        pw = self.packed_weights(dt)
        if single_stream:
            pw = pw.on_one_stream()
        kw = {}
        if isinstance(corr, PoolLookup):
            kw["pool"], kw["coords"] = (corr.levels, corr.slots, corr.num_slots), corr.coords
        else:
            kw["corr"] = cl(corr)
        if static_terms is not None:
            kw["P"] = static_terms
        else:
            kw["inp"] = cl(inp)
        agg = None
        if ii is not None:
            agg = agg_segments if agg_segments is not None else self._segments_from(ii)
        net, heads, eta, upmask = db.update_operator(pw, cl(net), cl(flow), agg=agg, **kw)
        hp = heads.permute(0, 2, 3, 1)                       # physical [E,H,W,8]: delta | delta_dy | weight | delta_mask
        batch, num = out_dim[0], out_dim[1]
        delta = hp[..., 0:4].reshape(batch, num, ht, wd, 4)
        weight = hp[..., 4:6].reshape(batch, num, ht, wd, 2)
        delta_m = hp[..., 6:8].reshape(batch, num, ht, wd, 2)
        net = net.view(*out_dim)
        if ii is None:
            return net, delta, weight, delta_m
        upmask = {"disp": upmask.view(batch, -1, 8 * 8 * 9, ht, wd), "flow": None, "dy_mask": None}
        return net, delta, weight, eta.view(batch, -1, ht, wd), upmask, delta_m

    def forward(self, net, inp, corr, flow=None, ii=None, jj=None, use_aff_bri=False, raw_mask=None, segments=None,
                agg_segments=None, static_terms=None, single_stream=False):
        """DynamicUpdateModule.forward (droid_net.py:256-314).  agg_segments (optional): (seg_ptr int32 [K+1], seg_idx
        int32 [E], K), the CSR of edges grouped by source frame; static_terms (optional): `self.static_terms(inp)` cached
        by the caller; corr may be a `PoolLookup` (16-bit inference only)."""
        batch, num, ch, ht, wd = net.shape
        if flow is None:
            flow = torch.zeros(batch, num, 4 + self.mask_num + 2, ht, wd, device=net.device, dtype=net.dtype)
        out_dim = (batch, num, -1, ht, wd)
        cl = torch.channels_last if net.is_cuda else torch.contiguous_format
        net = net.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)
        inp = inp.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl) if inp is not None else None
        pooled = isinstance(corr, PoolLookup)
        if not pooled:
            corr = corr.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)
        flow = flow.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)

        # 16-bit inference either through autocast (fp32 module) or with a module converted by .half()/.bfloat16()
        pdt = self.gru.convq.weight.dtype
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else pdt
        fused = (net.is_cuda and not self.training and not torch.is_grad_enabled() and self.fused_gru
                 and dt in (torch.float16, torch.bfloat16) and not use_aff_bri)
        if fused:
            return self._forward_fused(net, inp, corr, flow, ii, agg_segments, static_terms, dt, out_dim, single_stream)
        if pooled:
            raise RuntimeError("a PoolLookup needs the 16-bit inference path")
        if pdt != torch.float32 and not torch.is_autocast_enabled("cuda"):
            net, inp, flow, corr = (t.to(pdt) for t in (net, inp, flow, corr))
        corr = self.corr_encoder(corr)
        flow = self.flow_encoder(flow)
        net = self.gru(net, inp, corr, flow)
        delta, delta_dy, weight, delta_m = self._heads(net)
        if use_aff_bri:
            aff = self.param_linear(self.global_avg_pool(net).view(batch * num, -1)).view(batch, num, -1)

        to_last = lambda t: t.reshape(*out_dim).permute(0, 1, 3, 4, 2).contiguous()
        delta = torch.cat([to_last(delta), to_last(delta_dy)], dim=-1)      # droid_net.py:299
        weight, delta_m = to_last(weight), to_last(delta_m)
        net = net.view(*out_dim)

        if ii is None:
            return net, delta, weight, delta_m
        eta, upmask_disp, upmask_flow, upmask_dy = self.agg(net, ii.to(net.device))
        upmask = {"disp": upmask_disp, "flow": upmask_flow, "dy_mask": upmask_dy}
        if use_aff_bri:
            return net, delta, weight, eta, upmask, delta_m, aff
        return net, delta, weight, eta, upmask, delta_m
