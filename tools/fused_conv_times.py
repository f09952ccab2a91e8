"""MIOpen's fused conv+bias+ReLU (aten::miopen_convolution_relu) vs bias-free conv + pvo_bias_act"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pvo_amd import droid_backends as db
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
E, H, W = 36, 48, 64
cl = torch.channels_last
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, n, cin, cout, k in (("corr_enc0 1x1 196->128", E, 196, 128, 1), ("flow_enc0 7x7 8->128", E, 8, 128, 7), ("agg conv1 3x3 128->128", E, 128, 128, 3),
                              ("corr_enc2 3x3 128->128", E, 128, 128, 3)):
    x = torch.randn(n, cin, H, W, device=dev).half().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).half().contiguous(memory_format=cl)
    b = torch.randn(cout, device=dev)
    bh = b.half()
    p = k // 2
    with torch.no_grad():
        a = t(lambda: db.bias_act_(F.conv2d(x, w, None, padding=p).contiguous(memory_format=cl), b))
        try:
            f = lambda: torch.ops.aten.miopen_convolution_relu(x, w, bh, [1, 1], [p, p], [1, 1], 1)
            c = t(f)
            y1 = db.bias_act_(F.conv2d(x, w, None, padding=p).contiguous(memory_format=cl), b)
            y2 = f()
            d = (y1.float() - y2.float()).abs().max().item()
            fmt = y2.is_contiguous(memory_format=cl)
        except Exception as ex:
            c, d, fmt = float("nan"), str(ex)[:80], None
    print(f"{name:26s} conv+bias_act {a:7.1f} us   miopen_convolution_relu {c:7.1f} us   maxdiff {d}  channels_last_out={fmt}")
