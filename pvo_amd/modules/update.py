"""Update operator of the VO hot path: ConvGRU + heads + graph aggregation.

State-dict compatible with the reference's `DynamicUpdateModule`
(VO_Module/droid_slam/droid_net.py:166-314), `ConvGRU` (modules/gru.py:5-34), `GraphAgg`
(droid_net.py:64-95) and `GradientClip` (modules/clipping.py:7-23): same parameter names,
shapes and creation order, so a reference checkpoint's `update.*` tensors load unchanged
and a seeded default init reproduces the reference's weights.

What differs is how the forward pass is issued on MI355X (inference path):
  * the z and r gate convolutions read the same 448-channel input: one 256-output conv;
  * the four heads (delta, delta_dy, weight, delta_mask) share their input: one 512-output
    3x3 conv + ReLU, then one 3x3 conv with a block-diagonal weight producing the 8 outputs;
  * everything runs channels-last so MIOpen picks its NHWC implicit-GEMM (MFMA) kernels;
  * the element-wise half of the GRU (2 concats, gates, context mean, state blend: ~14 launches
    over 28-100 MB tensors in the reference formulation) is 4 hand-written HIP kernels around one
    persistent 448-channel buffer (pvo_amd/csrc/gru_fused.hip), under fp16/bf16 autocast.
The fused weights are views built from the individual parameters (cached in eval mode).
Training mode keeps the per-layer path so autograd sees the original parameters.
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


import os as _os
_HIP_CONV3 = _os.environ.get("PVO_HIP_CONV3") == "1"
_AGG_HIP_CONV = _os.environ.get("PVO_AGG_HIP_CONV", "1") == "1"
_FUSED_GRU_EPILOGUE = _os.environ.get("PVO_FUSED_GRU_EPILOGUE", "1") == "1"
_GRU_NO_ASSEMBLE = _os.environ.get("PVO_GRU_NO_ASSEMBLE", "0") == "1"   # measured: 125 vs 129 keyframe updates/s with the assembled input
_HIP_WIDE_CONV = _os.environ.get("PVO_HIP_WIDE_CONV", "1") == "1"     # GRU gate/candidate + heads' first stage on pvo_conv3x3
# The aggregation branch (conv1 over the edges, mean per source frame, then four small kernels over the K keyframes that
# leave most of the chip idle) runs on a second HIP stream beside the heads, which only share its input.  "0": one stream.
_AGG_SIDE_STREAM = _os.environ.get("PVO_AGG_SIDE_STREAM", "1") != "0"
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s



def _taps_wide(owner, key, weight_fn, dt):
    """[9,Cout,Cin] tap-major filter for pvo_conv3x3, cached on `owner` under `key`; weight_fn() returns [Cout,Cin,3,3]"""
    cache = owner.__dict__.setdefault("_taps_wide_cache", {})
    hit = cache.get(key)
    if hit is None or hit.dtype != dt:
        from .. import droid_backends as db
        hit = cache[key] = db.conv3x3_weights(weight_fn(), dt)
    return hit


def _w16(owner, conv, dt):
    """conv.weight in `dt`, channels-last, cached on `owner` (MIOpen's NHWC solvers want NHWC filters; without the
    cache PyTorch re-lays the filter out on every call - a 5 us copy kernel per convolution)."""
    cache = owner.__dict__.setdefault("_w16_cache", {})
    w = conv.weight
    hit = cache.get(id(conv))
    if hit is None or hit[0] is not w or hit[1] != w._version or hit[2].dtype != dt or hit[2].device != w.device:
        hit = cache[id(conv)] = (w, w._version, w.detach().to(dt).contiguous(memory_format=torch.channels_last))
    return hit[2]


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
        self._fb = None
        self._ws = None
        self._P_key = None
        self.__dict__.pop("_taps_wide_cache", None)
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

    def fused_forward(self, net, inp, corr_feat, flow_feat, corr_bias=None, flow_bias=None):
        """inference path on the fused HIP element-wise kernels (pvo_amd/csrc/gru_fused.hip).
        net, inp [E,128,H,W], corr_feat [E,128,H,W] and flow_feat [E,64,H,W] (both BEFORE their
        trailing ReLU), all channels-last and 16-bit.  Convolutions stay in MIOpen."""
        from .. import droid_backends as db
        E, c, h, w = net.shape
        dt = net.dtype
        wz, bz, wg, bg = self._fused_zr()
        key = (E, h, w, dt, net.device)
        if getattr(self, "_bufs_key", None) != key:
            mk = lambda ch: torch.empty(E, h, w, ch, dtype=dt, device=net.device).permute(0, 3, 1, 2)
            self._bufs, self._bufs_key = (mk(320), mk(128)), key
        X, Z = self._bufs
        fb = self._fused_bias()
        ws = self._split_weights(dt)
        # `inp` is constant over the life of an edge and convolution is linear in its input channels:
        # conv(W, [net|inp|corr|flow]) = conv(W[:, dyn], [net|corr|flow]) + conv(W[:, inp], inp).  The second term
        # is computed once per edge set and added inside the gate kernels: 128 of 448 input channels (29 %)
        # leave the two largest convolutions of every update.
        pk = (inp.data_ptr(), inp._version, tuple(inp.shape), dt)
        if getattr(self, "_P_key", None) != pk:
            self._P = (F.conv2d(inp, ws["zr_inp"], None, padding=1).contiguous(memory_format=torch.channels_last),
                       F.conv2d(inp, ws["q_inp"], None, padding=1).contiguous(memory_format=torch.channels_last))
            self._P_key = pk
        P_zr, P_q = self._P
        # every convolution below runs WITHOUT bias; the biases ride along in the fused kernels
        part = db.gru_glo_fused(net, _w16(self, self.w, dt), fb["w"])             # [E,K,128] partial means; 1x1 conv in the kernel
        K = part.shape[1]
        if fb.get("K") != K:                                                # gate weights tiled K times: the GEMM sums the chunks
            fb["wg_t_tiled"], fb["K"] = fb["wg_t"].repeat(K, 1).contiguous(), K
        with torch.autocast("cuda", enabled=False):
            g = torch.addmm(fb["g"], part.view(E, K * c), fb["wg_t_tiled"])   # context of z | r | q (+ conv biases), fp32
        if _HIP_WIDE_CONV and _FUSED_GRU_EPILOGUE and _GRU_NO_ASSEMBLE and corr_bias is not None and flow_bias is not None:
            # the two wide convolutions read net / corr features / flow features from their own tensors (bias + ReLU of the
            # features applied while the halo is staged) and carry the gate arithmetic: no concatenated X, no zr, no q
            Zg, RN = db.gru_gates(net, corr_feat, flow_feat, corr_bias, flow_bias,
                                  _taps_wide(self, "zr", lambda: ws["zr_dyn"], dt), g, P_zr)
            return db.gru_candidate(RN, corr_feat, flow_feat, corr_bias, flow_bias,
                                    _taps_wide(self, "q", lambda: ws["q_dyn"], dt), g, P_q, Zg, net)
        db.gru_assemble(net, None, corr_feat, flow_feat, X, corr_bias, flow_bias)   # X = [net | relu(cf) | relu(ff)]
        wide = _HIP_WIDE_CONV and X.shape[1] % 32 == 0
        if wide and _FUSED_GRU_EPILOGUE:
            # both large convolutions with the gate arithmetic as their epilogue: zr and q never reach HBM, r*net goes to
            # its own tensor (the candidate kernel reads [r*net | X[:, 128:]]), two element-wise kernels disappear
            Zg, RN = db.gru_conv_gates(X, _taps_wide(self, "zr", lambda: ws["zr_dyn"], dt), g, P_zr, net)
            return db.gru_conv_candidate(X, RN, _taps_wide(self, "q", lambda: ws["q_dyn"], dt), g, P_q, Zg, net)
        zr = db.conv3x3(X, _taps_wide(self, "zr", lambda: ws["zr_dyn"], dt)) if wide else F.conv2d(X, ws["zr_dyn"], None, padding=1)
        db.gru_gate(zr, g, net, Z, X, P_zr)                                 # X[:, :128] <- r * net
        q = db.conv3x3(X, _taps_wide(self, "q", lambda: ws["q_dyn"], dt)) if wide else F.conv2d(X, ws["q_dyn"], None, padding=1)
        return db.gru_out(q, g, Z, net, P_q)

    def _split_weights(self, dt):
        ws = getattr(self, "_ws", None)
        if ws is None or ws["q_dyn"].dtype != dt or ws["q_dyn"].device != self.convq.weight.device:
            wz = self._fused_zr()[0]
            dyn = lambda w: torch.cat([w[:, :128], w[:, 256:]], 1).to(dt).contiguous(memory_format=torch.channels_last)
            sta = lambda w: w[:, 128:256].to(dt).contiguous(memory_format=torch.channels_last)
            wq = self.convq.weight.detach()
            ws = self._ws = {"zr_dyn": dyn(wz), "zr_inp": sta(wz), "q_dyn": dyn(wq), "q_inp": sta(wq)}
        return ws

    def _fused_bias(self):
        fb = getattr(self, "_fb", None)
        if fb is None or fb["w"].device != self.w.weight.device:
            _, bz, wg, bg = self._fused_zr()
            c = self.w.weight.shape[0]
            # g = Wg glo + bg + [bz | br | bq]: the z/r/q convolution biases are per-channel constants too
            fb = self._fb = {"w": self.w.bias.detach().float().contiguous(),
                             "g": (bg.float() + torch.cat([bz.float(), self.convq.bias.detach().float()])).contiguous(),
                             "wg_t": wg.view(3 * c, c).float().t().contiguous()}
        return fb

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

    def forward(self, net, ii, segments=None, raw_eta=False):
        """raw_eta (fast path only): return (bias-free eta convolution output [K,1,H,W] 16-bit, its bias f32 [1]) in
        place of eta, for pvo_eta_finish (softplus, scaling and the damping bookkeeping in one kernel).
        segments (optional): (seg_ptr int32 [K+1], seg_idx int32 [E], K) — the CSR of edges grouped by source
        frame in the order of sorted(unique(ii)).  With it the grouping needs no torch.unique (which synchronises
        with the host to size its output) and the mean is one HIP kernel instead of zeros + 2 index_add + divide."""
        batch, num, ch, ht, wd = net.shape
        net = net.reshape(batch * num, ch, ht, wd)
        fast = (segments is not None and batch == 1 and net.is_cuda and not torch.is_grad_enabled()
                and net.dtype in (torch.float16, torch.bfloat16))
        if fast:
            from .. import droid_backends as db
            dt = net.dtype
            fb = self.__dict__.get("_fb32")
            if fb is None or fb[0].device != net.device:
                f32 = lambda t: t.detach().float().contiguous()
                fb = self.__dict__["_fb32"] = (f32(self.conv1.bias), f32(self.conv2.bias), f32(self.eta[0].bias))
            x = F.conv2d(net.contiguous(memory_format=torch.channels_last), _w16(self, self.conv1, dt), None, padding=1)
            # conv1's bias + ReLU are applied by the mean kernel as it reads (one pass over the 28 MB tensor instead of two)
            x = db.segment_mean(x.contiguous(memory_format=torch.channels_last), segments[0], segments[1], segments[2], in_bias=fb[0])
            if _AGG_HIP_CONV:   # K frames only: the hand-written 3x3 kernel (bias + ReLU inside) instead of MIOpen + a bias pass
                hit = self.__dict__.get("_taps2")
                w = self.conv2.weight
                if hit is None or hit[0] is not w or hit[1] != w._version or hit[2].dtype != dt or hit[2].device != w.device:
                    hit = self.__dict__["_taps2"] = (w, w._version, db.conv3x3_c128_weights(w, dt))
                net = db.conv3x3_c128(x, hit[2], fb[1], relu=True)
            else:
                net = F.conv2d(x, _w16(self, self.conv2, dt), None, padding=1)
                net = db.bias_act_(net.contiguous(memory_format=torch.channels_last), fb[1])
            # bias-free convolution (MIOpen adds a bias in a separate pass): eta's bias joins the fp32 softplus input
            eta_raw = F.conv2d(net, _w16(self, self.eta[0], dt), None, padding=1)
            # the 1x1 upsampling-mask layer over K frames as ONE GEMM with its bias ([K h w, 128] x [128, 576], hipBLASLt):
            # the grouped-convolution kernel MIOpen picks for this small batch takes 36 us and leaves the bias to a second pass
            up_w = self.__dict__.get("_up_gemm")
            w = self.upmask_disp[0].weight
            if up_w is None or up_w[0] is not w or up_w[1] != w._version or up_w[2].dtype != dt or up_w[2].device != w.device:
                up_w = self.__dict__["_up_gemm"] = (w, w._version, w.detach().reshape(w.shape[0], -1).t().to(dt).contiguous(),
                                                    self.upmask_disp[0].bias.detach().to(dt).contiguous())
            K = net.shape[0]
            x2 = net.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(K * ht * wd, net.shape[1])
            up = torch.addmm(up_w[3], x2, up_w[2]).view(K, ht, wd, -1).permute(0, 3, 1, 2)       # [K,576,h,w] channels-last
            upmask = up.view(batch, -1, 8 * 8 * 9, ht, wd)
            if raw_eta:
                return (eta_raw, fb[2]), upmask, None, None
            eta = F.softplus(eta_raw.float().add_(fb[2]))
            return eta.view(batch, -1, ht, wd).mul_(0.01), upmask, None, None
        else:
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
        self._b32 = None
        self._w2r = None
        self.fused_gru = True          # use pvo_amd/csrc/gru_fused.hip on the inference path
        # derived inference tensors (re-laid-out filters, fp32 biases, ...) are dropped whenever weights are (re)loaded
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_derived())

    def _drop_derived(self):
        for m in self.modules():
            for k in ("_taps_wide_cache", "_w16_cache", "_fb32", "_ftaps", "_enc0", "_taps3_cache", "_taps2", "_up_gemm"):
                m.__dict__.pop(k, None)
        self.train(self.training)

    def train(self, mode=True):
        self._fused_heads = None
        self._b32 = None
        self._w2r = None
        self.__dict__.pop("_taps_wide_cache", None)
        return super().train(mode)

    def _bias32(self):
        b = getattr(self, "_b32", None)
        if b is None or b["c0"].device != self.corr_encoder[0].bias.device:
            f = lambda t: t.detach().float().contiguous()
            b = self._b32 = {"c0": f(self.corr_encoder[0].bias), "c2": f(self.corr_encoder[2].bias),
                             "f0": f(self.flow_encoder[0].bias), "f2": f(self.flow_encoder[2].bias),
                             "h1": torch.cat([f(h[0].bias) for h in (self.delta, self.delta_dy, self.weight, self.delta_mask)]),
                             "h2": torch.cat([f(h[2].bias) for h in (self.delta, self.delta_dy, self.weight, self.delta_mask)]),
                             "a1": f(self.agg.conv1.bias), "a2": f(self.agg.conv2.bias)}
        return b

    def _taps3(self, conv, dt):
        cache = self.__dict__.setdefault("_taps3_cache", {})
        w = conv.weight
        hit = cache.get(id(conv))
        if hit is None or hit[0] is not w or hit[1] != w._version or hit[2].dtype != dt or hit[2].device != w.device:
            from .. import droid_backends as db
            hit = cache[id(conv)] = (w, w._version, db.conv3x3_c128_weights(w, dt))
        return hit[2]

    def _enc0_w(self, dt):
        w = self.corr_encoder[0].weight
        hit = self.__dict__.get("_enc0")
        if hit is None or hit[0] is not w or hit[1] != w._version or hit[2].dtype != dt or hit[2].device != w.device:
            from .. import droid_backends as db
            hit = self.__dict__["_enc0"] = (w, w._version, db.corr_encoder_weights(w, dt))
        return hit[2]

    def _flow_taps(self, dt):
        w = self.flow_encoder[0].weight
        hit = self.__dict__.get("_ftaps")
        if hit is None or hit[0] is not w or hit[1] != w._version or hit[2].dtype != dt or hit[2].device != w.device:
            from .. import droid_backends as db
            hit = self.__dict__["_ftaps"] = (w, w._version, db.conv7x7_c8_weights(w, dt))
        return hit[2]

    def _heads_w2(self, dt):
        w = getattr(self, "_w2r", None)
        if w is None or w.dtype != dt or w.device != self.delta[2].weight.device:
            hs = (self.delta, self.delta_dy, self.weight, self.delta_mask)
            # [head][out][tap = ky*3+kx][channel]
            w = self._w2r = torch.stack([h[2].weight.detach().permute(0, 2, 3, 1).reshape(2, 9, 128) for h in hs]).to(dt).contiguous()
        return w

    def _heads(self, net):
        """delta, delta_dy, weight, delta_mask, each [B,2,H,W]"""
        if self.training or torch.is_grad_enabled():
            return self.delta(net), self.delta_dy(net), self.weight(net), self.delta_mask(net)
        hs = (self.delta, self.delta_dy, self.weight, self.delta_mask)
        f = self._fused_heads
        if f is None or f[0].dtype != hs[0][0].weight.dtype or f[0].device != hs[0][0].weight.device:
            w1 = torch.cat([h[0].weight for h in hs], 0).contiguous(memory_format=torch.channels_last)
            b1 = torch.cat([h[0].bias for h in hs], 0)
            # second stage as ONE dense conv with a block-diagonal weight (4 x [2,128,3,3] on the
            # diagonal of [8,512,3,3]): a groups=4 conv with 2 outputs per group has no tuned
            # MIOpen solver and falls back to its naive kernel (110 ms per call in the profile)
            w2 = torch.zeros(8, 512, 3, 3, dtype=hs[0][2].weight.dtype, device=hs[0][2].weight.device)
            for k, h in enumerate(hs):
                w2[2 * k:2 * k + 2, 128 * k:128 * k + 128] = h[2].weight.detach()
            w2 = w2.contiguous(memory_format=torch.channels_last)
            b2 = torch.cat([h[2].bias for h in hs], 0)
            f = self._fused_heads = (w1.detach(), b1.detach(), w2.detach(), b2.detach())
        if net.is_cuda and net.dtype in (torch.float16, torch.bfloat16) and net.is_contiguous(memory_format=torch.channels_last):
            from .. import droid_backends as db
            # first stage 128 -> 4*128 as one bias-free MIOpen conv; bias, ReLU and the four 128 -> 2 second-stage
            # convolutions happen in ONE hand-written kernel (a 512 -> 8 conv has no efficient GEMM shape)
            if _HIP_WIDE_CONV:
                x = db.conv3x3(net, _taps_wide(self, "heads1", lambda: f[0], net.dtype))
            else:
                x = F.conv2d(net, f[0].to(net.dtype), None, padding=1).contiguous(memory_format=torch.channels_last)
            b32 = self._bias32()
            y = db.heads_out(x, b32["h1"], self._heads_w2(net.dtype), b32["h2"])
            self._last_heads = y                       # [E,8,H,W] channels-last: delta | delta_dy | weight | delta_mask
            return y[:, 0:2], y[:, 2:4], y[:, 4:6], y[:, 6:8]
        else:
            x = F.relu(F.conv2d(net, f[0], f[1], padding=1), inplace=True)  # 128 -> 4*128
        y = F.conv2d(x, f[2].to(x.dtype), f[3].to(x.dtype), padding=1)      # 4 x (128 -> 2), block diagonal
        return y[:, 0:2], y[:, 2:4], y[:, 4:6], y[:, 6:8]

    def forward(self, net, inp, corr, flow=None, ii=None, jj=None, use_aff_bri=False, raw_mask=None, segments=None,
                agg_segments=None, raw_heads=False):
        """raw_heads=True (inference on the HIP path): returns (net, heads [E,8,H,W] channels-last, eta, upmask) with the
        four head outputs side by side, for pvo_graph_post; raises if the fused head kernel was not used."""
        batch, num, ch, ht, wd = net.shape
        if flow is None:
            flow = torch.zeros(batch, num, 4 + self.mask_num + 2, ht, wd, device=net.device, dtype=net.dtype)
        out_dim = (batch, num, -1, ht, wd)
        cl = torch.channels_last if net.is_cuda else torch.contiguous_format
        net = net.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)
        inp = inp.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)
        if not callable(corr):
            corr = corr.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)
        flow = flow.reshape(batch * num, -1, ht, wd).contiguous(memory_format=cl)

        # 16-bit inference either through autocast (fp32 module) or with a module converted by .half()/.bfloat16()
        pdt = self.gru.convq.weight.dtype
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else pdt
        fused = (net.is_cuda and not self.training and not torch.is_grad_enabled() and self.fused_gru
                 and dt in (torch.float16, torch.bfloat16))
        if pdt != torch.float32 and not torch.is_autocast_enabled("cuda"):
            net, inp, flow = (t.to(pdt) for t in (net, inp, flow))
            corr = corr if callable(corr) else corr.to(pdt)
        if callable(corr) and not fused:
            raise RuntimeError("a fused lookup+encoder callable needs the 16-bit inference path")
        if fused:
            from .. import droid_backends as db
            cl_ = lambda t: t.to(dt).contiguous(memory_format=torch.channels_last)
            b32 = self._bias32()
            conv = lambda m, x, **kw: F.conv2d(x, _w16(self, m, dt), None, **kw)      # bias-free MIOpen convolution
            if callable(corr):        # a (coords-bound) fused lookup + first encoder layer: the 196 channels stay on chip
                c1 = corr(self._enc0_w(dt), b32["c0"])
            else:
                c1 = db.bias_act_(cl_(conv(self.corr_encoder[0], cl_(corr))), b32["c0"])             # + bias, ReLU: one pass
            f1 = db.conv7x7_c8(cl_(flow), self._flow_taps(dt), b32["f0"])     # 7x7, 8 -> 128, + bias + ReLU: one MFMA kernel
            if _HIP_CONV3:      # A/B switch: the hand-written 128-input 3x3 kernel instead of MIOpen (same speed)
                cf = db.conv3x3_c128(c1, self._taps3(self.corr_encoder[2], dt))
                ff = db.conv3x3_c128(f1, self._taps3(self.flow_encoder[2], dt))
            else:
                cf = conv(self.corr_encoder[2], c1, padding=1)              # their bias + ReLU happen in gru_assemble
                ff = conv(self.flow_encoder[2], f1, padding=1)
            net = self.gru.fused_forward(cl_(net), cl_(inp), cl_(cf), cl_(ff), b32["c2"], b32["f2"])
        else:
            corr = self.corr_encoder(corr)
            flow = self.flow_encoder(flow)
            net = self.gru(net, inp, corr, flow)

        self._last_heads = None
        if raw_heads:
            if ii is None:
                raise RuntimeError("raw_heads needs ii")
            run_agg = lambda: self.agg(net.view(*out_dim), ii.to(net.device), agg_segments, raw_eta=agg_segments is not None)
            if _AGG_SIDE_STREAM and net.is_cuda and not torch.cuda.is_current_stream_capturing():
                # fork / join around the two branches.  Everything the side stream allocates is either freed inside the
                # block or handed to the main stream after the join, and the next fork waits for the main stream again,
                # so the caching allocator never hands a block to one stream while the other can still touch it
                main, side = torch.cuda.current_stream(net.device), _side_stream(net.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    eta, upmask_disp, _, _ = run_agg()
                self._heads(net)
                main.wait_stream(side)
            else:
                self._heads(net)
                eta, upmask_disp, _, _ = run_agg()
            if self._last_heads is None:
                raise RuntimeError("raw_heads needs the fused 16-bit inference path")
            return net.view(*out_dim), self._last_heads, eta, {"disp": upmask_disp, "flow": None, "dy_mask": None}
        delta, delta_dy, weight, delta_m = self._heads(net)
        if use_aff_bri:
            aff = self.param_linear(self.global_avg_pool(net).view(batch * num, -1)).view(batch, num, -1)

        to_last = lambda t: t.reshape(*out_dim).permute(0, 1, 3, 4, 2).contiguous()
        delta = torch.cat([to_last(delta), to_last(delta_dy)], dim=-1)      # droid_net.py:299
        weight, delta_m = to_last(weight), to_last(delta_m)
        net = net.view(*out_dim)

        if ii is None:
            return net, delta, weight, delta_m
        eta, upmask_disp, upmask_flow, upmask_dy = self.agg(net, ii.to(net.device), agg_segments)
        upmask = {"disp": upmask_disp, "flow": upmask_flow, "dy_mask": upmask_dy}
        if use_aff_bri:
            return net, delta, weight, eta, upmask, delta_m, aff
        return net, delta, weight, eta, upmask, delta_m
