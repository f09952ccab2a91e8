import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
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
x = torch.randn(E, 128, H, W, device=dev).half().contiguous(memory_format=cl)
for co in (128, 512, 640):
    w = (torch.randn(co, 128, 3, 3, device=dev) * 0.03).half().contiguous(memory_format=cl)
    print("3x3 128 ->", co, "%.1f us" % t(lambda: F.conv2d(x, w, None, padding=1)))
x3 = torch.randn(E, 320, H, W, device=dev).half().contiguous(memory_format=cl)
for co in (128, 256, 384):
    w = (torch.randn(co, 320, 3, 3, device=dev) * 0.03).half().contiguous(memory_format=cl)
    print("3x3 320 ->", co, "%.1f us" % t(lambda: F.conv2d(x3, w, None, padding=1)))

from pvo_amd import droid_backends as db
for cin, co in ((320, 256), (320, 128), (128, 512), (128, 128)):
    xx = torch.randn(E, cin, H, W, device=dev).half().contiguous(memory_format=cl)
    w = (torch.randn(co, cin, 3, 3, device=dev) * 0.03).half()
    wt = db.conv3x3_weights(w, torch.half)
    wcl = w.contiguous(memory_format=cl)
    a = t(lambda: db.conv3x3(xx, wt)); b = t(lambda: F.conv2d(xx, wcl, None, padding=1))
    fl = 2 * E * H * W * cin * 9 * co
    print("3x3 %d -> %d: HIP %.1f us (%.2f PFLOP/s)   MIOpen %.1f us (%.2f PFLOP/s)" % (cin, co, a, fl / a / 1e9, b, fl / b / 1e9))
