import ctypes, sys, torch
dev = torch.device("cuda:0")
E_, H, W = 36, 48, 64
h1 = torch.randn(E_, H, W, 512, device=dev).half()
b1 = torch.randn(512, device=dev); b2 = torch.randn(8, device=dev)
w2 = (torch.randn(4, 2, 9, 128, device=dev) * 0.05).half()
y = torch.empty(E_, H, W, 8, device=dev, dtype=torch.half)
for name in sys.argv[1:]:
    lib = ctypes.CDLL(name)
    f = lib.pvo_heads_out; f.restype = ctypes.c_int
    vp = ctypes.c_void_p
    args = (vp(h1.data_ptr()), vp(b1.data_ptr()), vp(w2.data_ptr()), vp(b2.data_ptr()), vp(y.data_ptr()), E_, H, W, 1, vp(0))
    for _ in range(5): f(*args)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.current_stream().synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50): f(*args)
    torch.cuda.synchronize()
    if "ref" not in globals(): ref = y.clone()
    print(name.split("/")[-1], "%.1f us" % ((time.perf_counter() - t0) / 50 * 1e6), "max diff vs first %.4g" % (y.float() - ref.float()).abs().max().item())
