"""Encoders, upsampling helpers and the unrolled DroidNet.forward against outputs of the reference's
own droid_net.py / modules/extractor.py (tests/golden/gen_golden.py: gen_droidnet)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch


from pvo_amd import droid_net as dn
from pvo_amd.geom.se3 import SE3
from pvo_amd.modules.extractor import BasicEncoder

GOLD = os.path.join(os.path.dirname(__file__), "golden", "droidnet_forward.npz")


def _inputs(N, H, W, seed=12):
    """same stream as gen_golden.droidnet_inputs (kept in sync by test_inputs_helper_matches_generator)"""
    g = torch.Generator().manual_seed(seed)
    images = torch.randint(0, 256, (1, N, 3, H, W), generator=g).float()
    xi = torch.tensor([0.05, 0.01, 0.02, 0.003, 0.01, -0.004])
    Gs = SE3(torch.stack([SE3.exp(k * xi).data for k in range(N)], 0)[None])
    disps = 0.5 + 0.5 * torch.rand(1, N, H // 8, W // 8, generator=g)
    intr = torch.tensor([W * 0.1, W * 0.1, W / 16.0, H / 16.0])[None, None].repeat(1, N, 1)
    return images, Gs, disps, intr


def test_encoders_match_reference():
    z = np.load(GOLD)
    x = torch.from_numpy(z["enc_x"])
    for norm, od in (("instance", 128), ("none", 256)):
        torch.manual_seed(0)
        enc = BasicEncoder(output_dim=od, norm_fn=norm).eval()
        assert list(enc.state_dict().keys()) == list(z["enc_%s_keys" % norm])
        with torch.no_grad():
            y = enc(x)
        assert torch.allclose(y, torch.from_numpy(z["enc_%s" % norm]), atol=1e-5)


def test_upsampling_matches_reference():
    z = np.load(GOLD)
    up = dn.cvx_upsample(torch.from_numpy(z["cvx_data"]), torch.from_numpy(z["cvx_mask"]))
    assert torch.allclose(up, torch.from_numpy(z["cvx_up"]), atol=1e-6)
    assert torch.allclose(dn.upsample_inter(torch.from_numpy(z["inter_in"])), torch.from_numpy(z["inter_up"]), atol=1e-6)
    d = torch.from_numpy(z["cvx_data"])[..., 0][None]
    assert dn.upsample_dim_1(d, torch.from_numpy(z["cvx_mask"])[None]).shape == (1, 2, 40, 48)
    # convexity: a constant field stays constant
    c = dn.cvx_upsample(torch.full((1, 4, 4, 1), 2.5), torch.randn(1, 576, 4, 4))
    assert torch.allclose(c[:, 8:-8, 8:-8], torch.full_like(c[:, 8:-8, 8:-8], 2.5), atol=1e-6)


def test_state_dict_layout_matches_reference():
    z = np.load(GOLD)
    torch.manual_seed(0)
    net = dn.DroidNet()
    sd = net.state_dict()
    assert list(sd.keys()) == list(z["state_keys"])
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    assert np.allclose(sums, z["state_sums"], rtol=0, atol=1e-9)


class _OracleCorrBlock:
    """CPU stand-in for the HIP CorrBlock in the no-GPU test: torch volume + oracle lookup."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3):
        from oracle import oracle as O
        self.O, self.radius = O, radius
        from pvo_amd.modules.corr import CorrBlock
        self.pyr = CorrBlock._build_differentiable(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        b, n, h, w, _ = coords.shape
        c = coords.reshape(b * n, h, w, 2).permute(0, 3, 1, 2).contiguous().numpy()
        out = [torch.from_numpy(self.O.corr_index_forward(p.contiguous().numpy(), c / 2 ** i, self.radius)).view(b, n, -1, h, w)
               for i, p in enumerate(self.pyr)]
        return torch.cat(out, dim=2)


def _run_forward(device, corr_cls=None, atol=None):
    z = np.load(GOLD)
    N, H, W = [int(v) for v in z["shape"]]
    images, Gs, disps, intr = _inputs(N, H, W)
    torch.manual_seed(0)
    net = dn.DroidNet().eval().to(device)
    graph = OrderedDict((i, [j for j in range(N) if j != i and abs(i - j) <= 2]) for i in range(N))
    old = dn.CorrBlock
    if corr_cls is not None:
        dn.CorrBlock = corr_cls
    try:
        with torch.no_grad():
            res = net(SE3(Gs.data.clone().to(device)), images.to(device), disps.to(device), intr.to(device), graph,
                      num_steps=int(z["num_steps"]), fixedp=2, ret_flow=True, downsample=True)
    finally:
        dn.CorrBlock = old
    Gs_l, disp_l, resid_l, flow_l, mask_l = res
    for s in range(int(z["num_steps"])):
        assert torch.allclose(Gs_l[s].data[0].cpu(), torch.from_numpy(z["Gs_%d" % s]), atol=atol), s
        assert torch.allclose(disp_l[s][0, :, ::4, ::4].cpu(), torch.from_numpy(z["disp_up_%d" % s]), atol=atol), s
        assert torch.allclose(resid_l[s][0].cpu(), torch.from_numpy(z["resid_%d" % s]), atol=10 * atol), s
        assert torch.allclose(flow_l[s][0].cpu(), torch.from_numpy(z["flow_%d" % s]), atol=10 * atol), s
        assert torch.allclose(mask_l[s][0, :, ::8, ::8].cpu(), torch.from_numpy(z["mask_%d" % s]), atol=atol), s


def test_forward_matches_reference_cpu():
    _run_forward("cpu", _OracleCorrBlock, atol=1e-4)


@pytest.mark.gpu
def test_forward_matches_reference_gpu():
    """same unroll with the HIP correlation kernels and MIOpen convolutions (fp32)"""
    _run_forward("cuda:0", None, atol=2e-3)


@pytest.mark.gpu
def test_training_step_backpropagates_through_hip_lookup_and_ba():
    N, H, W = 3, 128, 128
    images, Gs, disps, intr = _inputs(N, H, W, seed=5)
    dev = "cuda:0"
    torch.manual_seed(0)
    net = dn.DroidNet().train().to(dev)
    graph = OrderedDict((i, [j for j in range(N) if j != i]) for i in range(N))
    res = net(SE3(Gs.data[:, :N].to(dev)), images.to(dev), disps.to(dev), intr.to(dev), graph, num_steps=2, fixedp=2)
    Gs_l, disp_l, resid_l, mask_l = res
    loss = sum(r.abs().mean() for r in resid_l) + sum(d.mean() for d in disp_l) + sum((g.data ** 2).sum() for g in Gs_l)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert any(k.startswith("fnet.") for k in grads), "no gradient reached the feature encoder through the lookup backward"
    assert any(k.startswith("update.gru.") for k in grads)
    assert all(torch.isfinite(g).all() for g in grads.values())
    assert sum(float(g.abs().sum()) for k, g in grads.items() if k.startswith("fnet.")) > 0


def _train_step(dev, corr_dtype, seed=5, N=3, H=128, W=128):
    images, Gs, disps, intr = _inputs(N, H, W, seed=seed)
    torch.manual_seed(0)
    net = dn.DroidNet().train().to(dev)
    graph = OrderedDict((i, [j for j in range(N) if j != i]) for i in range(N))
    res = net(SE3(Gs.data[:, :N].to(dev)), images.to(dev), disps.to(dev), intr.to(dev), graph, num_steps=2, fixedp=2,
              corr_dtype=corr_dtype)
    Gs_l, disp_l, resid_l, mask_l = res
    loss = sum(r.abs().mean() for r in resid_l) + sum(d.mean() for d in disp_l) + sum((g.data ** 2).sum() for g in Gs_l)
    loss.backward()
    return net, res, float(loss)


@pytest.mark.gpu
def test_bf16_volume_training_step_matches_fp32_volume():
    """BASELINE.json configs[4]: bf16 correlation volume (build, pyramid, HIP lookup forward + backward in bf16) with the
    update operator and the BA in fp32, against the same step with an fp32 volume: forward outputs to bf16 precision of
    the correlation features, gradients by direction (cosine) and size."""
    dev = "cuda:0"
    net32, res32, loss32 = _train_step(dev, None)
    net16, res16, loss16 = _train_step(dev, torch.bfloat16)
    assert abs(loss16 - loss32) < 2e-2 * abs(loss32)
    for a, b in zip(res16[0], res32[0]):                      # poses after each unrolled step
        assert (a.data - b.data).abs().max() < 5e-3
    for a, b in zip(res16[1], res32[1]):                      # upsampled inverse depths
        assert (a - b).abs().max() < 5e-2 and (a - b).abs().mean() < 2e-3
    g32 = {k: p.grad for k, p in net32.named_parameters() if p.grad is not None}
    g16 = {k: p.grad for k, p in net16.named_parameters() if p.grad is not None}
    assert set(g16) == set(g32)
    for group in ("fnet.", "cnet.", "update.gru.", "update.corr_encoder."):
        a = torch.cat([g16[k].flatten() for k in sorted(g16) if k.startswith(group)])
        b = torch.cat([g32[k].flatten() for k in sorted(g32) if k.startswith(group)])
        assert torch.isfinite(a).all() and b.abs().sum() > 0
        cos = torch.dot(a, b) / (a.norm() * b.norm())
        assert cos > 0.98, (group, float(cos))
        assert 0.8 < float(a.norm() / b.norm()) < 1.25, group


class _TorchCorrBlock:
    """differentiable CPU stand-in for the HIP CorrBlock (bilinear window sampling with grid_sample), used only to drive
    DDP's gradient plumbing on machines without a GPU"""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3):
        from pvo_amd.modules.corr import CorrBlock
        self.pyr, self.r = CorrBlock._build_differentiable(fmap1, fmap2, num_levels), radius

    def __call__(self, coords):
        import torch.nn.functional as F
        b, n, h, w, _ = coords.shape
        r = self.r
        d = torch.arange(-r, r + 1, dtype=coords.dtype)
        dx, dy = torch.meshgrid(d, d, indexing="ij")                          # channel = x-offset major (correlation_kernels.cu)
        out = []
        for i, p in enumerate(self.pyr):
            hl, wl = p.shape[-2:]
            c = coords.reshape(b * n * h * w, 1, 1, 2) / 2 ** i
            pts = c + torch.stack([dx, dy], -1).view(1, 2 * r + 1, 2 * r + 1, 2)
            gx = 2 * pts[..., 0] / max(wl - 1, 1) - 1
            gy = 2 * pts[..., 1] / max(hl - 1, 1) - 1
            s = F.grid_sample(p.reshape(b * n * h * w, 1, hl, wl), torch.stack([gx, gy], -1), align_corners=True)
            out.append(s.view(b, n, h, w, -1).permute(0, 1, 4, 2, 3))
        return torch.cat(out, dim=2)


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    old = dn.CorrBlock
    dn.CorrBlock = _TorchCorrBlock
    try:
        N, H, W = 3, 64, 64
        torch.manual_seed(0)
        net = dn.DroidNet().train()
        ddp = DDP(net, find_unused_parameters=False)             # train.py:64
        graph = OrderedDict((i, [j for j in range(N) if j != i]) for i in range(N))

        def loss_of(model, seed):
            images, Gs, disps, intr = _inputs(N, H, W, seed=seed)
            Gs_l, disp_l, resid_l, flow_l, mask_l = model(SE3(Gs.data[:, :N]), images, disps, intr, graph, num_steps=1, fixedp=2,
                                                          ret_flow=True, downsample=True)
            # every head takes part (the reference's losses cover flow, depth, residual and mask terms, train.py:178-261),
            # so find_unused_parameters can stay False as in train.py:64
            return (sum(r.abs().mean() for r in resid_l) + sum(d.mean() for d in disp_l) + sum(f.abs().mean() for f in flow_l)
                    + sum(m.mean() for m in mask_l))
        loss_of(ddp, 20 + rank).backward()                       # every rank its own batch (DistributedSampler, train.py:87-88)
        g = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
        # what the average over both ranks' batches should be, computed without DDP
        torch.manual_seed(0)
        ref = dn.DroidNet().train()
        for s in range(world):
            (loss_of(ref, 20 + s) / world).backward()
        gr = torch.cat([p.grad.flatten() for p in ref.parameters() if p.grad is not None])
        out[rank] = (g.clone(), gr.clone())
    finally:
        dn.CorrBlock = old
        dist.destroy_process_group()


def test_ddp_two_ranks_average_gradients_cpu():
    """train.py:33-41,64: DroidNet under DistributedDataParallel, two gloo ranks with different batches - the all-reduced
    gradients are identical on both ranks and equal the average of the per-batch gradients"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, 29533, out), nprocs=2, join=True)
    (g0, r0), (g1, r1) = out[0], out[1]
    assert torch.equal(g0, g1)
    assert torch.allclose(g0, r0, atol=1e-6, rtol=1e-4)


@pytest.mark.gpu
def test_ddp_wraps_the_hip_training_step_on_rccl():
    """one-rank RCCL process group on cuda:0: DDP's bucketed all-reduce hooks fire through a backward pass that contains
    the HIP lookup backward (the 4-GPU run of configs[4] differs only in world size)"""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29534"
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        N, H, W = 3, 128, 128
        images, Gs, disps, intr = _inputs(N, H, W, seed=5)
        torch.manual_seed(0)
        net = dn.DroidNet().train().to(dev)
        ddp = DDP(net, device_ids=[0])
        graph = OrderedDict((i, [j for j in range(N) if j != i]) for i in range(N))
        Gs_l, disp_l, resid_l, flow_l, mask_l = ddp(SE3(Gs.data[:, :N].to(dev)), images.to(dev), disps.to(dev), intr.to(dev), graph,
                                                    num_steps=2, fixedp=2, ret_flow=True, downsample=True, corr_dtype=torch.bfloat16)
        (sum(r.abs().mean() for r in resid_l) + sum(d.mean() for d in disp_l) + sum(f.abs().mean() for f in flow_l)
         + sum(m.mean() for m in mask_l)).backward()
        torch.cuda.synchronize()
        n_grad = 0
        for k, p in net.named_parameters():
            assert p.grad is not None, k                        # every parameter's hook fired: the bucket was all-reduced
            assert torch.isfinite(p.grad).all(), k
            n_grad += 1
        assert n_grad == 110
    finally:
        dist.destroy_process_group()
