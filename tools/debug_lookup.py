import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from pvo_amd import droid_backends as db
cuda = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = 36, 48, 64
pyr = [torch.randn(N, H, W, H >> l, W >> l, generator=g).half() for l in range(4)]
base = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float()
coords = base[None] + torch.randn(N, H, W, 2, generator=g) * 4
got = db.corr_pyramid_lookup([p.to(cuda) for p in pyr], coords.to(cuda), 3).cpu()
sel = [0, 7, 20, 35]
want = O.corr_pyramid_lookup([p[sel].numpy() for p in pyr], coords[sel].numpy(), 3)
a = got[sel].numpy().view(np.uint16); b = want.view(np.uint16)
bad = np.argwhere(a != b)
print("mismatches", len(bad), "of", a.size)
groups = {}
for n, ch, y, x in bad:
    groups.setdefault((n, ch // 49, y, x), []).append(ch % 49)
print("pixel-level groups", len(groups))
for k, v in list(groups.items())[:8]:
    n, l, y, x = k
    print(k, "channels", len(v), "coords", coords[sel[n], y, x].tolist(), "ex got/want", got[sel][n, l*49+v[0], y, x].item(), want[n, l*49+v[0], y, x])
