"""time each convolution of the update operator (fp16, channels-last, S-B shapes) on the GPU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pvo_amd.modules.update import DynamicUpdateModule
torch.backends.cudnn.benchmark = "--nobench" not in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = DynamicUpdateModule().to(dev).eval().half()
E, K, H, W = 36, 8, 48, 64
cl = torch.channels_last
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = lambda c, n=E: torch.randn(n, c, H, W, device=dev, dtype=torch.half).contiguous(memory_format=cl)
xn = lambda c, n=E: torch.randn(n, c, H, W, device=dev, dtype=torch.half)
with torch.no_grad():
    rows = [
        ("corr_enc 1x1 196->128", lambda a=x(196): m.corr_encoder[0](a), 196*128),
        ("corr_enc 3x3 128->128", lambda a=x(128): m.corr_encoder[2](a), 128*128*9),
        ("flow_enc 7x7 8->128", lambda a=x(8): m.flow_encoder[0](a), 8*128*49),
        ("flow_enc 3x3 128->64", lambda a=x(128): m.flow_encoder[2](a), 128*64*9),
        ("gru w 1x1 128->128", lambda a=x(128): m.gru.w(a), 128*128),
        ("gru zr 3x3 448->256 (fused)", lambda a=x(448): F.conv2d(a, *m.gru._fused_zr()[:2], padding=1), 448*256*9),
        ("gru q 3x3 448->128", lambda a=x(448): m.gru.convq(a), 448*128*9),
        ("gru zr 3x3 320->256 (dyn split)", lambda a=x(320), w=torch.randn(256, 320, 3, 3, device=dev, dtype=torch.half).contiguous(memory_format=cl): F.conv2d(a, w, None, padding=1), 320*256*9),
        ("gru q 3x3 320->128 (dyn split)", lambda a=x(320), w=torch.randn(128, 320, 3, 3, device=dev, dtype=torch.half).contiguous(memory_format=cl): F.conv2d(a, w, None, padding=1), 320*128*9),
        ("heads 3x3 128->512 (fused)", lambda a=x(128): F.conv2d(a, torch.cat([h[0].weight for h in (m.delta, m.delta_dy, m.weight, m.delta_mask)]).contiguous(memory_format=cl), None, padding=1), 128*512*9),
        ("heads 3x3 512->8 blockdiag", lambda a=x(512): F.conv2d(a, torch.zeros(8, 512, 3, 3, device=dev, dtype=torch.half).contiguous(memory_format=cl), None, padding=1), 512*8*9),
        ("agg conv1 3x3 128->128", lambda a=x(128): m.agg.conv1(a), 128*128*9),
        ("agg conv2 3x3 128->128 K=8 NCHW", lambda a=xn(128, K): m.agg.conv2(a), 128*128*9),
        ("agg conv2 3x3 128->128 K=8 NHWC", lambda a=x(128, K): m.agg.conv2(a), 128*128*9),
        ("agg eta 3x3 128->1 K=8", lambda a=x(128, K): m.agg.eta[0](a), 128*9),
        ("agg upmask 1x1 128->576 K=8", lambda a=x(128, K): m.agg.upmask_disp[0](a), 128*576),
    ]
    tot = 0
    for name, fn, macs in rows:
        us = t(fn)
        n = K if "K=8" in name else E
        print(f"{name:38s} {us:9.1f} us   {2*macs*n*H*W/us/1e6:8.1f} TFLOP/s")
        tot += us
    print("sum", tot)
    net, inp, corr, motn = x(128)[None], x(128)[None], x(196)[None], x(8)[None]
    ii = torch.arange(K, device=dev).repeat_interleave(5)[:E].contiguous()
    ii = torch.cat([ii, torch.full((E - ii.numel(),), K - 1, device=dev)]) if ii.numel() < E else ii
    with torch.autocast("cuda", dtype=torch.float16):
        us = t(lambda: m(net, inp, corr, motn, ii, None))
    print(f"whole update operator: {us:.1f} us  ({0.59e12/us/1e6:.1f} TFLOP/s)")
