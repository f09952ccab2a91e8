"""Which layer of the per-frame encoders does not repeat itself?  Every piece of BasicEncoder.forward_inference is evaluated three times on
the same input and its bytes hashed; with `deterministic` as argument the vendor convolutions run under torch.backends.cudnn.flags(deterministic=True).
python tools/encoder_determinism.py [deterministic]   (GPU box)"""
import os, sys, hashlib, contextlib
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvo_amd.modules.extractor import BasicEncoder, _conv, _b
from pvo_amd import droid_backends as db

dev = torch.device("cuda:0")
det = len(sys.argv) > 1 and sys.argv[1] == "deterministic"
ctx = (lambda: torch.backends.cudnn.flags(enabled=True, deterministic=True, benchmark=False)) if det else contextlib.nullcontext
h = lambda t: hashlib.md5(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:8]
same = lambda xs: "same" if len(set(xs)) == 1 else "DIFFERENT " + str(xs)
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
img = (torch.rand(3, 240, 808, generator=g) * 255).int().to(dev)
bad = 0
with torch.no_grad(), ctx():
    for norm_fn, dim in (("instance", 128), ("none", 256)):
        net = BasicEncoder(output_dim=dim, norm_fn=norm_fn).to(dev).half().eval()
        norm = norm_fn == "instance"
        x = db.frame_normalise(img, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
        act = lambda t, bias, residual, nrm, ri, ro, out=None: db.bias_norm_act(t.contiguous(), bias, residual, norm=nrm, eps=1e-5, relu_inner=ri, relu_outer=ro, out=out)
        r = same([h(_conv(net.conv1, x)) for _ in range(3)]); bad += r != "same"
        print("%s encoder: stem 7x7 stride 2: %s" % (norm_fn, r))
        t = act(_conv(net.conv1, x), _b(net.conv1, x), None, norm, True, False)
        for li, layer in enumerate((net.layer1, net.layer2, net.layer3)):
            for bi, block in enumerate(layer):
                r1 = same([h(_conv(block.conv1, t)) for _ in range(3)])
                y = act(_conv(block.conv1, t), _b(block.conv1, t), None, norm, True, False)
                r2 = same([h(_conv(block.conv2, y)) for _ in range(3)])
                y2 = _conv(block.conv2, y)
                xx = t
                if block.downsample is not None:
                    d = block.downsample[0]
                    xx = act(db.conv1x1_planes(t.contiguous(), d.weight, _b(d, t), stride=d.stride[0]), None, None, norm, False, False)
                t = act(y2, _b(block.conv2, xx), xx, norm, True, True)
                bad += (r1 != "same") + (r2 != "same")
                print("  layer%d block%d: conv1 (%d -> %d, stride %d) %s | conv2 %s" % (li + 1, bi, block.conv1.in_channels, block.conv1.out_channels, block.conv1.stride[0], r1, r2))
        outs = [h(net.forward_inference(x[None])) for _ in range(3)]
        r = same(outs); bad += r != "same"
        print("  whole encoder, three calls: %s" % r)
print("vendor convolutions under cudnn.flags(deterministic=True): %s; layers that did not repeat: %d" % (det, bad))
