"""Back-to-back timings of the wide-layer convolution kernel at the update operator's shapes (S-B: 36 x 48 x 64).
    python tools/conv_bench.py [E H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvo_amd import droid_backends as db, _lib
if os.environ.get("PVO_LIB_PATH"):
    _lib.LIB_PATH = os.environ["PVO_LIB_PATH"]          # a probe / experiment build of the library

dev = torch.device("cuda:0")
E, H, W = 36, 48, 64
if len(sys.argv) > 3:
    E, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cl = torch.channels_last


def t(fn, n=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("E,H,W =", E, H, W)
for cin, cout in ((128, 128), (128, 256), (128, 512), (320, 128), (320, 256)):
    x = torch.randn(E, cin, H, W, device=dev).half().contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.02).half()
    wt = db.conv3x3_weights(w, torch.half)
    us = t(lambda: db.conv3x3(x, wt))
    fl = 2.0 * E * H * W * 9 * cin * cout
    print("conv3x3 %3d -> %3d : %7.1f us  %6.1f TFLOP/s  (%.3f of 2.5 PF)" % (cin, cout, us, fl / us / 1e6, fl / us / 1e6 / 2500))
    if cin == 128 and cout in (64, 128, 256, 512):
        wt2 = db.conv3x3_c128_weights(w, torch.half)
        us2 = t(lambda: db.conv3x3_c128(x, wt2))
        print("   c128 kernel       : %7.1f us  %6.1f TFLOP/s" % (us2, fl / us2 / 1e6))
net = torch.tanh(torch.randn(E, 128, H, W, device=dev)).half().contiguous(memory_format=cl)
cf = torch.relu(torch.randn(E, 192, H, W, device=dev)).half().contiguous(memory_format=cl)
gg = torch.randn(E, 384, device=dev)
P_zr = torch.randn(E, 256, H, W, device=dev).half().contiguous(memory_format=cl)
P_q = torch.randn(E, 128, H, W, device=dev).half().contiguous(memory_format=cl)
tzr = db.conv3x3_weights((torch.randn(256, 320, 3, 3, device=dev) * 0.02).half(), torch.half)
tq = db.conv3x3_weights((torch.randn(128, 320, 3, 3, device=dev) * 0.02).half(), torch.half)
Z, RN = db.gru_conv_gates(net, cf, tzr, gg, P_zr)
us = t(lambda: db.gru_conv_gates(net, cf, tzr, gg, P_zr)); fl = 2.0 * E * H * W * 9 * 320 * 256
print("gru_conv_gates      : %7.1f us  %6.1f TFLOP/s  (%.3f)" % (us, fl / us / 1e6, fl / us / 1e6 / 2500))
us = t(lambda: db.gru_conv_candidate(RN, cf, tq, gg, P_q, Z, net)); fl = 2.0 * E * H * W * 9 * 320 * 128
print("gru_conv_candidate  : %7.1f us  %6.1f TFLOP/s  (%.3f)" % (us, fl / us / 1e6, fl / us / 1e6 / 2500))
