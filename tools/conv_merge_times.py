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
