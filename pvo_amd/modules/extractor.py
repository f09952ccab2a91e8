"""Feature / context encoders (1/8 resolution), state-dict compatible with the reference's
``modules/extractor.py`` (VO_Module/droid_slam/modules/extractor.py:6-56 ResidualBlock,
:116-201 BasicEncoder): same sub-module names, creation order and initialisation (kaiming-normal
fan_out for convs, unit/zero for norm affine parameters :166-173), so a reference checkpoint loads
with `load_state_dict` and a seeded construction gives identical weights.

These run once per frame (MotionFilter) and are vendor-library convolutions (MIOpen); they are
caller-side plumbing of the hot path, not a hand-written kernel.

Round 5: `BasicEncoder.forward_inference` is the same network for 16-bit inference on the GPU with everything BETWEEN the
convolutions - bias, instance norm, ReLU, the residual add - as one kernel per layer (`pvo_bias_norm_act`): 36 launches instead
of ~95 per network and frame (the full-sequence run of bench.py spends a tracked frame's 0.9 of 1.4 ms in those small kernels).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

DIM = 32


def _norm(kind, planes, groups=None):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups if groups is not None else planes // 8, num_channels=planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "none":
        return nn.Sequential()
    raise ValueError("unknown norm_fn %r" % (kind,))


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)

    def forward_inference(self, x, norm, act):
        """the same block on a 16-bit NCHW tensor: convolutions without their bias, `act` (droid_backends.bias_norm_act) for the rest"""
        y = act(_conv(self.conv1, x), _b(self.conv1, x), None, norm, True, False)
        y2 = _conv(self.conv2, y)
        if self.downsample is not None:
            d = self.downsample[0]
            from .. import droid_backends as db
            if d.kernel_size == (1, 1) and d.padding == (0, 0) and d.stride[0] == d.stride[1] and d.groups == 1 \
                    and db.conv1x1_planes_supported(d.in_channels, d.out_channels):
                # the strided shortcut on the library's 1 x 1 kernel (bias included; the vendor library: a sub-tensor copy, two layout
                # transposes and an implicit GEMM, ~30 us for 25 M multiply-adds)
                w = d.weight if d.weight.dtype == x.dtype else d.weight.to(x.dtype)
                x = act(db.conv1x1_planes(x.contiguous(), w, _b(d, x), stride=d.stride[0]), None, None, norm, False, False)
            else:
                x = act(_conv(d, x), _b(d, x), None, norm, False, False)
        return act(y2, _b(self.conv2, x), x, norm, True, True, out=y2)          # relu(x + relu(norm2(conv2(y))))


def _conv(m, x):
    w = m.weight if m.weight.dtype == x.dtype else m.weight.to(x.dtype)
    return F.conv2d(x, w, None, m.stride, m.padding)


def _b(m, x):
    return None if m.bias is None else (m.bias if m.bias.dtype == x.dtype else m.bias.to(x.dtype))


class BasicEncoder(nn.Module):
    """[B,N,3,H,W] -> [B,N,output_dim,H/8,W/8]  (extractor.py:183-201)."""
    deterministic = True         # forward_inference: vendor convolutions restricted to kernels that repeat their result (see there)

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0, multidim=False):
        super().__init__()
        if multidim:
            raise NotImplementedError("multidim encoders are never constructed on the VO path (droid_net.py:320-321)")
        self.norm_fn = norm_fn
        self.multidim = multidim
        self.norm1 = _norm(norm_fn, DIM, groups=8)
        self.conv1 = nn.Conv2d(3, DIM, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = DIM
        self.layer1 = self._make_layer(DIM, stride=1)
        self.layer2 = self._make_layer(2 * DIM, stride=2)
        self.layer3 = self._make_layer(4 * DIM, stride=2)
        self.conv2 = nn.Conv2d(4 * DIM, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        blocks = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                  ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*blocks)

    def forward(self, x):
        b, n, c, h, w = x.shape
        x = self.relu1(self.norm1(self.conv1(x.reshape(b * n, c, h, w))))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        return x.view(b, n, *x.shape[1:])

    def forward_inference(self, x, dtype=torch.float16):
        """forward() for 16-bit inference on the GPU (what fp16 autocast computes: every convolution, norm and add in `dtype`),
        the work between the convolutions fused per layer.  Falls back to forward() where that does not apply."""
        if not x.is_cuda or torch.is_grad_enabled() or self.training or self.norm_fn not in ("instance", "none") or self.dropout is not None:
            return self.forward(x)
        from .. import droid_backends as db
        # The vendor library picks a convolution kernel per shape from timings taken on the box, and some of its candidates add split-K
        # partial sums with atomics: on some boxes the 128 -> 128 3 x 3 layers at 1/8 resolution came out different on every call
        # (tools/encoder_determinism.py), and with them every run of a sequence.  `deterministic` restricts the choice to kernels that
        # repeat themselves; it is this process's global flag, so it is set for the duration of the call only.
        prev = torch.backends.cudnn.deterministic
        torch.backends.cudnn.deterministic = bool(self.deterministic) or prev
        try:
            return self._forward_inference(x, dtype, db)
        finally:
            torch.backends.cudnn.deterministic = prev

    def _forward_inference(self, x, dtype, db):
        norm = self.norm_fn == "instance"
        eps = 1e-5

        def act(t, bias, residual, nrm, relu_in, relu_out, out=None):
            return db.bias_norm_act(t.contiguous(), bias, residual, norm=nrm, eps=eps, relu_inner=relu_in, relu_outer=relu_out,
                                    out=out if out is not None and out.is_contiguous() else None)
        b, n, c, h, w = x.shape
        t = x.reshape(b * n, c, h, w).to(dtype)
        t = act(_conv(self.conv1, t), _b(self.conv1, t), None, norm, True, False)
        for layer in (self.layer1, self.layer2, self.layer3):
            for block in layer:
                t = block.forward_inference(t, norm, act)
        c2 = self.conv2
        if c2.kernel_size == (1, 1) and c2.stride == (1, 1) and c2.padding == (0, 0) and c2.groups == 1 \
                and db.conv1x1_planes_supported(c2.in_channels, c2.out_channels):
            # the last layer on the library's own kernel, bias included: the vendor library's 1x1 convolution of this shape adds split-K
            # partial sums with atomics - the same frame gave a different feature map on every call (tools/determinism_probe.py)
            w = c2.weight if c2.weight.dtype == t.dtype else c2.weight.to(t.dtype)
            t = db.conv1x1_planes(t.contiguous(), w, _b(c2, t))
        else:
            t = act(_conv(c2, t), _b(c2, t), None, False, False, False)
        return t.view(b, n, *t.shape[1:])
