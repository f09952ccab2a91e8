"""time pvo_ba (1 iteration) from ablation builds of ba.hip: python tools/ba_probe.py lib1.so lib2.so ..."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_geom_ba_gpu import _scene
dev = torch.device("cuda:0")
nf = int(os.environ.get("NF", "8"))
s = _scene(0, nf, 48, 64, 3, 1)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
E, F, H, W = d["ii"].shape[0], d["disps"].shape[0], 48, 64
vp = ctypes.c_void_p
for name in sys.argv[1:]:
    lib = ctypes.CDLL(name)
    lib.pvo_ba_workspace_bytes.restype = ctypes.c_size_t
    nbytes = lib.pvo_ba_workspace_bytes(E, nf - 1, F, H * W)
    ws = torch.empty(nbytes + 512, dtype=torch.uint8, device=dev)
    eta = d["eta"].contiguous().view(-1, H * W)
    poses0, disps0 = d["poses"].clone(), d["disps"].clone()
    poses, disps = poses0.clone(), disps0.clone()
    def run():
        return lib.pvo_ba(vp(poses.data_ptr()), vp(disps.data_ptr()), vp(d["intr"].data_ptr()), vp(d["target"].data_ptr()),
                          vp(d["weight"].data_ptr()), vp(eta.data_ptr()), vp(d["ii"].data_ptr()), vp(d["jj"].data_ptr()),
                          E, F, H, W, eta.shape[0], 1, nf, 1, ctypes.c_float(1e-4), ctypes.c_float(0.1), 0,
                          vp(0), vp(0), 0, vp(0), vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()), vp(0))
    for _ in range(5):
        assert run() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    print(os.path.basename(name), "P=%d" % (nf - 1), "%.1f us per BA iteration (plan + 5 kernels)" % ((time.perf_counter() - t0) / 100 * 1e6))
    if hasattr(lib, "pvo_probe_ts"):
        import numpy as np
        ts = np.zeros(256, dtype=np.int64)
        lib.pvo_probe_ts(ctypes.c_void_p(ts.ctypes.data))
        P = nf - 1
        print("clock64 ticks (100 MHz => 10 ns each?) entry->loaded %d, factor total %d, back+out %d" % (ts[200] - ts[210], ts[201] - ts[200], ts[211] - ts[201]))
        for kb in range(P):
            a = ts[0] if kb == 0 else ts[4 + (kb - 1) * 4]
            print("  block %d: chol6 %d  panel %d  barrier %d  trailing+barrier %d" % (kb, ts[1 + kb * 4] - a, ts[2 + kb * 4] - ts[1 + kb * 4], ts[3 + kb * 4] - ts[2 + kb * 4], ts[4 + kb * 4] - ts[3 + kb * 4]))
