import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvo_amd import droid_backends as db
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for C in (128, 512):
    x = torch.randn(36, C, 48, 64, device=dev).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev)
    mb = x.numel() * 2 / 1e6
    us = t(lambda: db.bias_act_(x, b))
    print(f"bias_act C={C}: {us:.1f} us  ({2*mb/us/1e6*1e6/1e6:.2f} TB/s r+w of {mb:.0f} MB)")
    us = t(lambda: x.relu_())
    print(f"  torch relu_: {us:.1f} us ({2*mb/us:.2f} MB/us)")
    y = torch.empty_like(x)
    us = t(lambda: y.copy_(x))
    print(f"  torch copy: {us:.1f} us ({2*mb/us:.2f} MB/us)")
