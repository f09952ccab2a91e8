"""1x1 convolutions as plain GEMMs on the channels-last pixel matrix: hipBLASLt vs the MIOpen convolution"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
E, K8, H, W = 36, 8, 48, 64
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, n, cin, cout in (("corr_enc 1x1", E, 196, 128), ("gru w 1x1", E, 128, 128), ("upmask 1x1", K8, 128, 576)):
    x = torch.randn(n, H, W, cin, device=dev, dtype=torch.half)
    w = torch.randn(cout, cin, device=dev, dtype=torch.half) * 0.05
    b = torch.randn(cout, device=dev, dtype=torch.half)
    xc = x.permute(0, 3, 1, 2)
    wc = w.view(cout, cin, 1, 1).contiguous(memory_format=torch.channels_last)
    M = n * H * W
    xm = x.view(M, cin)
    wt = w.t().contiguous()
    with torch.no_grad():
        r = [("conv2d", t(lambda: F.conv2d(xc, wc, None))),
             ("conv2d+bias", t(lambda: F.conv2d(xc, wc, b))),
             ("linear", t(lambda: F.linear(xm, w))),
             ("linear+bias", t(lambda: F.linear(xm, w, b))),
             ("addmm", t(lambda: torch.addmm(b, xm, wt))),
             ("addmm_relu", t(lambda: torch._addmm_activation(b, xm, wt, use_gelu=False)))]
        y1 = torch.relu(F.conv2d(xc, wc, b)).permute(0, 2, 3, 1).reshape(M, cout)
        y2 = torch._addmm_activation(b, xm, wt, use_gelu=False)
    print(f"{name:14s} M={M} K={cin} N={cout}: " + "  ".join(f"{k} {v:.1f}us" for k, v in r), " maxdiff", (y1.float() - y2.float()).abs().max().item())
