"""back-to-back timings of the hand-written update-operator kernels at S-B (E=36, 48x64, fp16)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvo_amd import droid_backends as db
dev = torch.device("cuda:0")
E, H, W = 36, 48, 64
cl = torch.channels_last
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = lambda c: torch.randn(E, c, H, W, device=dev).half().contiguous(memory_format=cl)
b = lambda c: torch.randn(c, device=dev)
motn, net, h1 = x(8), torch.tanh(x(128)), x(512)
wt = db.conv7x7_c8_weights(torch.randn(128, 8, 7, 7, device=dev) * 0.05, torch.half)
ww = (torch.randn(128, 128, device=dev) * 0.1).half()
w2 = (torch.randn(4, 2, 9, 128, device=dev) * 0.05).half()
big = x(128)
rows = [("conv7x7_c8 (28 MB out)", lambda: db.conv7x7_c8(motn, wt, b(128)) if False else db.conv7x7_c8(motn, wt, B128)),
        ("gru_glo_fused (28 MB in)", lambda: db.gru_glo_fused(net, ww, B128)),
        ("heads_out (113 MB in)", lambda: db.heads_out(h1, B512, w2, B8)),
        ("bias_act_ (28+28 MB)", lambda: db.bias_act_(big, B128)),
        ("copy 28 MB (reference point)", lambda: big.clone())]
B128, B512, B8 = b(128), b(512), b(8)
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
for co in (64, 128, 256, 512):
    wc = (torch.randn(co, 128, 3, 3, device=dev) * 0.03).half()
    wcl = wc.contiguous(memory_format=cl)
    wt3 = db.conv3x3_c128_weights(wc, torch.half)
    rows.append(("conv3x3_c128 -> %d (HIP)" % co, lambda wt3=wt3: db.conv3x3_c128(big, wt3)))
    rows.append(("conv3x3 128 -> %d (MIOpen)" % co, lambda wcl=wcl: F.conv2d(big, wcl, None, padding=1)))
for name, fn in rows:
    print(f"{name:32s} {t(fn):7.1f} us")
