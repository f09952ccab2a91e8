"""Generate golden fixtures from the REFERENCE's own Python, run in the build container.

    python tests/golden/gen_golden.py          (needs /root/reference; never runs on the GPU box)

What is executed is the reference's code, imported from /root/reference/VO_Module/droid_slam:
  geom/ba.py (BA, MoBA), geom/chol.py, geom/projective_ops.py      -> ba_python_*.npz, projective_*.npz
  modules/corr.py  CorrBlock.corr + pyramid construction            -> corr_volume.npz
  modules/gru.py, droid_net.py (DynamicUpdateModule sub-modules, GraphAgg) -> update_op.npz
  geom/graph_utils.py graph_to_edge_list                            -> graph_edges.npz
  factor_graph.py FactorGraph.update (with recorded callees)        -> factor_graph_glue_*.npz
  geom/losses.py (every loss train.py combines, SSIM, unsupervised labels) -> train_losses.npz

Three native dependencies of that Python cannot be built in this image (lietorch's C++
extension needs Eigen/Core, which the vendored Eigen lacks; torch_scatter and the
droid_backends CUDA extension are absent).  They are substituted at import time:
  lietorch       -> pvo_amd.geom.se3.SE3 (the product's SE3, itself pinned by lietorch's
                    property tests, tests/test_se3.py)
  torch_scatter  -> scatter_sum / scatter_mean written with index_add_
  droid_backends -> an empty module (corr.py:4 imports it; nothing here calls it)
Only DATA is written: inputs, outputs and seeds.  No reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/VO_Module/droid_slam"
sys.path.insert(0, ROOT)


def install_substitutes():
    from pvo_amd.geom import se3 as myse3

    lt = types.ModuleType("lietorch")
    lt.SE3 = myse3.SE3

    class _Absent:  # Sim3 / SO3 are imported by name but never instantiated on this path
        pass
    lt.Sim3 = type("Sim3", (_Absent,), {})
    lt.SO3 = type("SO3", (_Absent,), {})
    lt.cat = myse3.cat
    lt.stack = myse3.stack
    sys.modules["lietorch"] = lt

    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        dim = dim % src.dim()
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        shape = list(src.shape)
        shape[dim] = dim_size
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        return res.index_add_(dim, index.to(src.device), src)

    def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
        s = scatter_sum(src, index, dim, None, dim_size)
        cnt = torch.zeros(s.shape[dim % src.dim()], dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index.to(src.device), torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
        shape = [1] * s.dim()
        shape[dim % src.dim()] = -1
        return s / cnt.clamp(min=1).view(shape)

    ts.scatter_sum, ts.scatter_mean = scatter_sum, scatter_mean
    sys.modules["torch_scatter"] = ts
    sys.modules["droid_backends"] = types.ModuleType("droid_backends")
    for m in ("cv2", "matplotlib", "matplotlib.pyplot"):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                sys.modules[m] = types.ModuleType(m)
    sys.path.insert(0, REF)


def make_scene(seed, P=5, ht=8, wd=10, radius=2, noise=0.05):
    """A small synthetic window: smooth depth > 0.25 everywhere, gentle motion."""
    g = torch.Generator().manual_seed(seed)
    from pvo_amd.geom.se3 import SE3
    intr = torch.tensor([wd * 0.8, wd * 0.8, wd / 2.0, ht / 2.0])
    xi = torch.tensor([0.06, 0.01, 0.03, 0.004, 0.012, -0.006])
    poses_gt = torch.stack([SE3.exp(k * xi).data for k in range(P)], 0)
    low = torch.rand(1, 1, 3, 4, generator=g) * 0.8 + 0.4
    disps_gt = torch.nn.functional.interpolate(low, size=(ht, wd), mode="bilinear", align_corners=True)[0, 0]
    disps_gt = disps_gt[None].repeat(P, 1, 1) * (1.0 + 0.05 * torch.arange(P).view(P, 1, 1))
    ii, jj = [], []
    for i in range(P):
        for j in range(P):
            if i != j and abs(i - j) <= radius:
                ii.append(i); jj.append(j)
    ii, jj = torch.tensor(ii), torch.tensor(jj)
    import geom.projective_ops as pops
    coords_gt, _ = pops.projective_transform(SE3(poses_gt[None]), disps_gt[None], intr[None, None].repeat(1, P, 1), ii, jj)
    E = ii.shape[0]
    target = coords_gt[0] + noise * torch.randn(E, ht, wd, 2, generator=g)
    weight = torch.rand(E, ht, wd, 2, generator=g)
    # initial state: poses lag one frame behind, depth is flat
    poses0 = torch.stack([SE3.exp(max(k - 1, 0) * xi).data if k > 0 else poses_gt[0] for k in range(P)], 0)
    poses0[1] = SE3.exp(0.5 * xi).data
    disps0 = torch.full((P, ht, wd), 0.7)
    eta = 0.05 + 0.1 * torch.rand(P, ht, wd, generator=g)
    return dict(intr=intr, poses=poses0, disps=disps0, target=target, weight=weight, eta=eta, ii=ii, jj=jj)


def gen_ba():
    from pvo_amd.geom.se3 import SE3
    from geom.ba import BA, MoBA
    import geom.projective_ops as pops
    for name, seed, P, ht, wd, fixedp in [("a", 0, 5, 8, 10, 1), ("b", 1, 4, 6, 9, 2)]:
        s = make_scene(seed, P, ht, wd)
        intr_all = s["intr"][None, None].repeat(1, P, 1)
        out = {k: v.numpy() for k, v in s.items()}
        out["fixedp"] = np.int64(fixedp)
        # one reprojection with Jacobians (pins coords/valid and, through BA, the Jacobians)
        coords, valid = pops.projective_transform(SE3(s["poses"][None]), s["disps"][None], intr_all, s["ii"], s["jj"])
        out["reproj_coords"] = coords[0].numpy(); out["reproj_valid"] = valid[0].numpy()
        # back-projection of every pixel through its frame's pose (what droid_backends.iproj computes: pose.act((X,Y,1,d)) / w),
        # written with the reference's pops.iproj
        X0, _ = pops.iproj(s["disps"][None], intr_all)
        Xw = SE3(s["poses"][None])[:, :, None, None] * X0
        out["iproj_points"] = (Xw[0, ..., :3] / Xw[0, ..., 3:]).numpy()
        # mean magnitude of the flow induced by camera motion, per edge: frame_distance's beta = 1 term
        flow, _ = pops.induced_flow(SE3(s["poses"][None]), s["disps"][None], intr_all, s["ii"], s["jj"])
        out["induced_flow_mean"] = flow[0].norm(dim=-1).mean(dim=(1, 2)).numpy()
        Gs, disps = SE3(s["poses"][None].clone()), s["disps"][None].clone()
        for it in range(2):
            # the reference adds 1e-7 to C on top of eta (ba.py:91); the native path does not
            Gs, disps = BA(s["target"][None], s["weight"][None], s["eta"][None] - 1e-7, Gs, disps,
                           intr_all, s["ii"], s["jj"], fixedp=fixedp)
            out["ba_poses_%d" % (it + 1)] = Gs.data[0].numpy().copy()
            out["ba_disps_%d" % (it + 1)] = disps[0].numpy().copy()
        Gm = MoBA(s["target"][None], s["weight"][None], None, SE3(s["poses"][None].clone()), s["disps"][None],
                  intr_all, s["ii"], s["jj"], fixedp=fixedp)
        out["moba_poses_1"] = Gm.data[0].numpy().copy()
        # training path: gradients of a fixed linear functional of two BA steps (chol.py:22-30 backward)
        g = torch.Generator().manual_seed(100 + seed)
        leaves = {k: s[k][None].clone().requires_grad_(True) for k in ("target", "weight", "eta", "disps")}
        Gs, disps = SE3(s["poses"][None].clone()), leaves["disps"]
        for it in range(2):
            Gs, disps = BA(leaves["target"], leaves["weight"], leaves["eta"] - 1e-7, Gs, disps,
                           intr_all, s["ii"], s["jj"], fixedp=fixedp)
        cp, cd = torch.randn(Gs.data.shape, generator=g), torch.randn(disps.shape, generator=g)
        ((Gs.data * cp).sum() + (disps * cd).sum()).backward()
        out["grad_cp"], out["grad_cd"] = cp[0].numpy(), cd[0].numpy()
        for k, v in leaves.items():
            out["grad_" + k] = v.grad[0].numpy().copy()
        np.savez_compressed(os.path.join(HERE, "ba_python_%s.npz" % name), **out)
        print("ba_python_%s: P=%d E=%d %dx%d" % (name, P, s["ii"].shape[0], ht, wd))


def gen_corr():
    from modules.corr import CorrBlock
    g = torch.Generator().manual_seed(2)
    N, C, H, W = 1, 32, 16, 16   # the reference pools once more than it keeps (corr.py:35-38): H,W >= 16
    f1 = torch.randn(1, N, C, H, W, generator=g)
    f2 = torch.randn(1, N, C, H, W, generator=g)
    out = dict(fmap1=f1[0].numpy(), fmap2=f2[0].numpy())
    cb = CorrBlock(f1, f2, num_levels=4, radius=3)
    for l, lv in enumerate(cb.corr_pyramid):
        out["level%d_f32" % l] = lv.numpy()
    cbh = CorrBlock(f1.half(), f2.half(), num_levels=4, radius=3)
    for l, lv in enumerate(cbh.corr_pyramid):
        out["level%d_f16" % l] = lv.numpy()
    np.savez_compressed(os.path.join(HERE, "corr_volume.npz"), **out)
    print("corr_volume: N=%d C=%d %dx%d" % (N, C, H, W))


def gen_update_op():
    """DynamicUpdateModule, executed sub-module by sub-module in the order of
    droid_net.py:276-303 (its forward() itself raises on np.range at :295)."""
    import droid_net
    torch.manual_seed(0)
    upd = droid_net.DynamicUpdateModule()
    upd.eval()
    g = torch.Generator().manual_seed(3)
    E, ht, wd = 3, 6, 8
    net = torch.tanh(torch.randn(1, E, 128, ht, wd, generator=g))
    inp = torch.relu(torch.randn(1, E, 128, ht, wd, generator=g))
    corr = torch.randn(1, E, 196, ht, wd, generator=g)
    motion = torch.randn(1, E, 8, ht, wd, generator=g).clamp(-64, 64)
    ii = torch.tensor([0, 0, 1])
    with torch.no_grad():
        n_ = net.view(E, -1, ht, wd); i_ = inp.view(E, -1, ht, wd)
        c_ = upd.corr_encoder(corr.view(E, -1, ht, wd))
        f_ = upd.flow_encoder(motion.view(E, -1, ht, wd))
        n_ = upd.gru(n_, i_, c_, f_)
        delta = upd.delta(n_); delta_dy = upd.delta_dy(n_); weight = upd.weight(n_); delta_m = upd.delta_mask(n_)
        eta, upmask, _, _ = upd.agg(n_.view(1, E, 128, ht, wd), ii)
    sd = upd.state_dict()
    keys = sorted(sd.keys())
    digest = np.array([float(sd[k].double().sum()) for k in keys])
    np.savez_compressed(os.path.join(HERE, "update_op.npz"),
                        net=net.numpy(), inp=inp.numpy(), corr=corr.numpy(), motion=motion.numpy(), ii=ii.numpy(),
                        out_net=n_.numpy(), out_delta=delta.numpy(), out_delta_dy=delta_dy.numpy(),
                        out_weight=weight.numpy(), out_delta_m=delta_m.numpy(), out_eta=eta.numpy(),
                        out_upmask=upmask.numpy(), state_keys=np.array(keys), state_sums=digest,
                        state_shapes=np.array([str(tuple(sd[k].shape)) for k in keys]))
    print("update_op: %d state tensors, %d params" % (len(keys), sum(v.numel() for v in sd.values())))


def gen_graph():
    """FactorGraph.update glue (factor_graph.py:227-307) with a recorded update operator:
    inputs/outputs of the arithmetic between the update operator and the BA call."""
    import geom.graph_utils as gu
    from collections import OrderedDict
    graph = OrderedDict()
    for i in range(4):
        graph[i] = [j for j in range(4) if i != j and abs(i - j) <= 2]
    ii, jj, kk = gu.graph_to_edge_list(graph)
    np.savez_compressed(os.path.join(HERE, "graph_edges.npz"), ii=ii.numpy(), jj=jj.numpy(), kk=kk.numpy())
    print("graph_edges: E=%d" % ii.shape[0])


def gen_factor_graph_glue():
    """FactorGraph.update (factor_graph.py:227-307) with recorded stand-ins for the three things
    it calls (video.reproject, the update operator, video.ba): pins the arithmetic in between —
    motion features, mask / weight / damping glue, the panoptic segment vote, the BA arguments."""
    import factor_graph as ref_fg
    g = torch.Generator().manual_seed(5)
    E, ht, wd, F = 6, 6, 8, 4
    ii = torch.tensor([0, 0, 1, 1, 2, 3]); jj = torch.tensor([1, 2, 0, 2, 3, 2])
    rec = dict(
        coords1=torch.randn(1, E, ht, wd, 2, generator=g) * 3 + 4,
        target_cam=torch.randn(1, E, ht, wd, 2, generator=g) * 3 + 4,
        weight0=torch.rand(1, E, ht, wd, 2, generator=g),
        raw_mask=torch.randn(1, E, ht, wd, 2, generator=g),
        delta_dy=torch.randn(1, E, ht, wd, 2, generator=g) * 0.2,
        net=torch.randn(1, E, 128, ht, wd, generator=g), inp=torch.randn(1, E, 128, ht, wd, generator=g),
        corr=torch.randn(1, E, 196, ht, wd, generator=g),
        net_out=torch.randn(1, E, 128, ht, wd, generator=g),
        delta=torch.randn(1, E, ht, wd, 4, generator=g) * 0.5,
        weight_out=torch.randn(1, E, ht, wd, 2, generator=g),
        damping_out=torch.rand(1, 4, ht, wd, generator=g) * 0.01,
        delta_m=torch.randn(1, E, ht, wd, 2, generator=g),
        segm=torch.randint(0, 4, (1, E, 1, ht, wd), generator=g).int(),
    )
    for segm_filter in (False, True):
        cap = {}

        class Video:
            pass
        v = Video()
        v.ht, v.wd = ht * 8, wd * 8
        v.disps = torch.ones(F, ht, wd)
        v.segm_filter, v.thresh = segm_filter, 0.5
        v.reproject = lambda a, b: (rec["coords1"].clone(), torch.ones(1, E, ht, wd, 1))

        def ba(target, weight, eta, ii_, jj_, t0, t1, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
            cap.update(ba_target=target.clone(), ba_weight=weight.clone(), ba_eta=eta.clone(), ba_ii=ii_.clone(),
                       ba_jj=jj_.clone(), ba_t0=torch.tensor(t0), ba_itrs=torch.tensor(itrs))
        v.ba = ba

        def update_op(net, inp, corr, motn, ii_, jj_, flag):
            cap["motn"] = motn.clone()
            return rec["net_out"], rec["delta"], rec["weight_out"], rec["damping_out"], {}, rec["delta_m"]
        fg = ref_fg.FactorGraph(v, update_op, device="cpu")
        fg.ii, fg.jj, fg.age = ii.clone(), jj.clone(), torch.zeros(E, dtype=torch.long)
        fg.net, fg.inp, fg.segm = rec["net"].clone(), rec["inp"].clone(), rec["segm"].clone()
        fg.target_cam, fg.weight = rec["target_cam"].clone(), rec["weight0"].clone()
        fg.raw_mask, fg.delta_dy = rec["raw_mask"].clone(), rec["delta_dy"].clone()
        fg.corr = lambda c: rec["corr"]
        fg.update(None, 4, itrs=2)
        out = {k: t.numpy() for k, t in rec.items() if k not in ("net", "inp", "corr", "net_out")}   # opaque to the glue
        out.update({k: t.numpy() for k, t in cap.items()})
        out.update(ii=ii.numpy(), jj=jj.numpy(), out_target_cam=fg.target_cam.numpy(), out_weight=fg.weight.numpy(),
                   out_raw_mask=fg.raw_mask.numpy(), out_delta_dy=fg.delta_dy.numpy(), out_full_flow=fg.full_flow.numpy(),
                   out_damping=fg.damping.numpy(), out_age=fg.age.numpy())
        np.savez_compressed(os.path.join(HERE, "factor_graph_glue_%s.npz" % ("segm" if segm_filter else "plain")), **out)
        print("factor_graph_glue segm_filter=%s" % segm_filter)


def _oracle_backed_lookup():
    """droid_backends.corr_index_forward/backward for the reference's CorrSampler at fixture time:
    the CPU oracle (oracle/oracle_corr.c), fp32."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    mod = sys.modules["droid_backends"]

    def fwd(volume, coords, radius):
        return [torch.from_numpy(O.corr_index_forward(volume.detach().numpy(), coords.detach().numpy(), radius))]

    def bwd(volume, coords, grad, radius):
        return [torch.from_numpy(O.corr_index_backward(tuple(volume.shape), coords.detach().numpy(),
                                                       grad.detach().numpy(), radius))]
    mod.corr_index_forward, mod.corr_index_backward = fwd, bwd


def droidnet_inputs(N, H, W, seed=12):
    """Inputs of the DroidNet.forward fixture, regenerated from the seed by the generator and the tests
    alike (CPU generator streams are reproducible), so the images need not be stored."""
    from pvo_amd.geom.se3 import SE3
    g = torch.Generator().manual_seed(seed)
    images = torch.randint(0, 256, (1, N, 3, H, W), generator=g).float()
    xi = torch.tensor([0.05, 0.01, 0.02, 0.003, 0.01, -0.004])
    Gs = SE3(torch.stack([SE3.exp(k * xi).data for k in range(N)], 0)[None])
    disps = 0.5 + 0.5 * torch.rand(1, N, H // 8, W // 8, generator=g)
    intr = torch.tensor([W * 0.1, W * 0.1, W / 16.0, H / 16.0])[None, None].repeat(1, N, 1)
    return images, Gs, disps, intr


def gen_droidnet():
    """The reference's DroidNet.forward (droid_net.py:342-439), run unmodified on a 4-frame graph.
    Fixture-time substitutes on top of install_substitutes(): the lookup extension is the CPU oracle, and
    numpy gets a `range` attribute aliasing `arange` so the dead line droid_net.py:295 (whose result is never
    used) does not raise.  Weights are the seeded default initialisation; only inputs/outputs are stored."""
    from collections import OrderedDict
    import droid_net as ref_net
    from modules.extractor import BasicEncoder
    from pvo_amd.geom.se3 import SE3
    _oracle_backed_lookup()
    np.range = np.arange
    out = {}
    # encoders: seeded construction, one small input
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 2, 3, 32, 48, generator=g)
    for norm, od in (("instance", 128), ("none", 256)):
        torch.manual_seed(0)
        enc = BasicEncoder(output_dim=od, norm_fn=norm).eval()
        out["enc_%s" % norm] = enc(x).detach().numpy()
        out["enc_%s_keys" % norm] = np.array(list(enc.state_dict().keys()))
    out["enc_x"] = x.numpy()
    # convex upsampling
    data, mask = torch.randn(2, 5, 6, 3, generator=g), torch.randn(2, 576, 5, 6, generator=g)
    out["cvx_data"], out["cvx_mask"] = data.numpy(), mask.numpy()
    out["cvx_up"] = ref_net.cvx_upsample(data, mask).numpy()
    fl = torch.randn(1, 2, 4, 5, 2, generator=g)
    out["inter_in"], out["inter_up"] = fl.numpy(), ref_net.upsample_inter(fl).numpy()
    # the unrolled loop
    torch.manual_seed(0)
    net = ref_net.DroidNet().eval()
    N, H, W = 4, 128, 160
    images, Gs, disps, intr = droidnet_inputs(N, H, W)
    graph = OrderedDict((i, [j for j in range(N) if j != i and abs(i - j) <= 2]) for i in range(N))
    with torch.no_grad():
        res = net(SE3(Gs.data.clone()), images.clone(), disps.clone(), intr, graph, num_steps=3, fixedp=2, ret_flow=True,
                  downsample=True)
    Gs_l, disp_l, resid_l, flow_l, mask_l = res
    out.update(shape=np.array([N, H, W]), num_steps=np.int64(3), state_keys=np.array(list(net.state_dict().keys())),
               state_sums=np.array([float(v.double().sum()) for v in net.state_dict().values()]))
    for s in range(3):
        out["Gs_%d" % s] = Gs_l[s].data[0].numpy()
        out["disp_up_%d" % s] = disp_l[s][0, :, ::4, ::4].numpy()
        out["resid_%d" % s] = resid_l[s][0].numpy()
        out["flow_%d" % s] = flow_l[s][0].numpy()
        out["mask_%d" % s] = mask_l[s][0, :, ::8, ::8].numpy()
    np.savez_compressed(os.path.join(HERE, "droidnet_forward.npz"), **out)
    print("droidnet_forward: %d frames %dx%d, %d state tensors" % (N, H, W, len(net.state_dict())))


LOWMEM_EDGES = ([0, 0, 1, 5, 8, 8, 9, 9], [1, 2, 0, 6, 9, 7, 8, 7])


def lowmem_records(seed=6):
    """recorded stand-in outputs for the update_lowmem fixture (shared by the generator and the test)"""
    g = torch.Generator().manual_seed(seed)
    E, ht, wd, F = 8, 5, 7, 10
    return dict(
        E=E, ht=ht, wd=wd, F=F,
        coords1=[torch.randn(1, E, ht, wd, 2, generator=g) * 3 + 4 for _ in range(2)],
        target_cam=torch.randn(1, E, ht, wd, 2, generator=g) * 3 + 4,
        weight0=torch.rand(1, E, ht, wd, 2, generator=g),
        raw_mask=torch.randn(1, E, ht, wd, 2, generator=g),
        delta_dy=torch.randn(1, E, ht, wd, 2, generator=g) * 0.2,
        net=torch.randn(1, E, 128, ht, wd, generator=g),
        net_out=[torch.randn(1, E, 128, ht, wd, generator=g) for _ in range(2)],
        delta=[torch.randn(1, E, ht, wd, 4, generator=g) * 0.5 for _ in range(2)],
        weight_out=[torch.randn(1, E, ht, wd, 2, generator=g) for _ in range(2)],
        damp_table=[torch.rand(F, ht, wd, generator=g) * 0.01 for _ in range(2)],
        delta_m=[torch.randn(1, E, ht, wd, 2, generator=g) for _ in range(2)],
        inps=torch.randn(F, 128, ht, wd, generator=g), fmaps=torch.randn(F, 128, ht, wd, generator=g),
    )


def gen_lowmem_glue():
    """FactorGraph.update_lowmem (factor_graph.py:309-360), two steps over two source-frame chunks, with recorded
    stand-ins for video.reproject, the alt-corr block, the update operator and video.ba: pins the chunking, the motion
    features (built from target_cam - coords0, not the fresh reprojection), the raw-eta damping and the BA arguments
    (t0 = 1, t1 = counter, lm = 1e-5, ep = 1e-2)."""
    import factor_graph as ref_fg
    rec = lowmem_records()
    E, ht, wd, F = rec["E"], rec["ht"], rec["wd"], rec["F"]
    ii, jj = torch.tensor(LOWMEM_EDGES[0]), torch.tensor(LOWMEM_EDGES[1])
    cap = {"motn": [], "ba": [], "calls": []}
    step = {"k": -1}

    class Counter:
        value = F

    class Video:
        pass
    v = Video()
    v.ht, v.wd, v.counter = ht * 8, wd * 8, Counter()
    v.disps, v.dirty = torch.ones(F, ht, wd), torch.zeros(F, dtype=torch.bool)
    v.segm_filter, v.thresh = False, 0.5
    v.fmaps, v.inps = rec["fmaps"], rec["inps"]

    def reproject(a, b):
        step["k"] += 1
        return rec["coords1"][step["k"]].clone(), torch.ones(1, E, ht, wd, 1)
    v.reproject = reproject

    def ba(target, weight, eta, ii_, jj_, t0, t1, itrs=2, lm=1e-4, ep=0.1, motion_only=False):
        cap["ba"].append(dict(target=target.clone(), weight=weight.clone(), eta=eta.clone(), t0=t0, t1=t1, itrs=itrs, lm=lm, ep=ep,
                              motion_only=motion_only))
    v.ba = ba

    class FakeAlt:
        def __init__(self, fmaps, *a, **k):
            pass

        def __call__(self, coords, ii_, jj_):
            return torch.zeros(1, ii_.shape[0], 196, ht, wd)
    ref_fg.AltCorrBlock = FakeAlt

    def update_op(net, inp, corr, motn, ii_, jj_, flag):
        k = step["k"]
        sel = torch.tensor([e for e in range(E) if (int(ii[e]) // 8) == (int(ii_[0]) // 8)])
        cap["motn"].append(motn.clone()); cap["calls"].append(sel.clone())
        frames = torch.unique(ii_)
        return (rec["net_out"][k][:, sel], rec["delta"][k][:, sel], rec["weight_out"][k][:, sel],
                rec["damp_table"][k][frames][None], {}, rec["delta_m"][k][:, sel])
    fg = ref_fg.FactorGraph(v, update_op, device="cpu", corr_impl="alt")
    fg.ii, fg.jj, fg.age = ii.clone(), jj.clone(), torch.zeros(E, dtype=torch.long)
    fg.net = rec["net"].clone()
    fg.target_cam, fg.weight = rec["target_cam"].clone(), rec["weight0"].clone()
    fg.raw_mask, fg.delta_dy = rec["raw_mask"].clone(), rec["delta_dy"].clone()
    fg.update_lowmem(steps=2)
    out = dict(ii=ii.numpy(), jj=jj.numpy(), n_calls=np.int64(len(cap["calls"])),
               out_target_cam=fg.target_cam.numpy(), out_weight=fg.weight.numpy(), out_raw_mask=fg.raw_mask.numpy(),
               out_delta_dy=fg.delta_dy.numpy(), out_damping=fg.damping.numpy(), out_net=fg.net.numpy().astype(np.float16),
               dirty=v.dirty.numpy())
    for n, m in enumerate(cap["motn"]):
        out["motn_%d" % n] = m.numpy(); out["call_edges_%d" % n] = cap["calls"][n].numpy()
    for n, b in enumerate(cap["ba"]):
        out["ba_target_%d" % n] = b["target"].numpy(); out["ba_weight_%d" % n] = b["weight"].numpy(); out["ba_eta_%d" % n] = b["eta"].numpy()
        out["ba_args_%d" % n] = np.array([b["t0"], b["t1"], b["itrs"], b["lm"], b["ep"], float(b["motion_only"])])
    np.savez_compressed(os.path.join(HERE, "lowmem_glue.npz"), **out)
    print("lowmem_glue: %d operator calls, %d BA calls" % (len(cap["calls"]), len(cap["ba"])))


def proximity_case(seed=8):
    g = torch.Generator().manual_seed(seed)
    t = 12
    d = torch.rand(t, t, generator=g) * 40.0                       # pairwise "frame distances", some above the thresholds
    d = 0.5 * (d + d.t())
    have = ([0, 1, 1, 2, 5, 9, 9], [1, 0, 2, 1, 9, 5, 4])            # active edges, two of them long range
    bad = ([7], [2])
    inac = ([3, 8], [0, 3])
    return t, d, have, bad, inac


def gen_proximity():
    """FactorGraph.add_proximity_factors (factor_graph.py:372-429) with a recorded distance matrix: pins the greedy
    selection (radius edges, distance threshold, non-maximum suppression around existing / chosen edges) and the order
    in which edges are handed to add_factors."""
    import factor_graph as ref_fg
    t, d, have, bad, inac = proximity_case()
    out = dict(d=d.numpy())

    class Counter:
        value = t

    class Video:
        pass
    v = Video()
    v.ht, v.wd, v.counter = 64, 64, Counter()
    v.disps = torch.ones(t, 8, 8)
    v.segm_filter = False
    v.distance = lambda ii, jj, beta=0.3: d[ii.long(), jj.long()].clone()
    for n, kw in enumerate((dict(t0=0, t1=0, rad=2, nms=2, thresh=16.0), dict(t0=7, t1=0, rad=2, nms=1, thresh=12.0),
                            dict(t0=2, t1=1, rad=1, nms=3, thresh=30.0))):
        fg = ref_fg.FactorGraph(v, None, device="cpu")
        fg.ii, fg.jj = torch.tensor(have[0]), torch.tensor(have[1])
        fg.ii_bad, fg.jj_bad = torch.tensor(bad[0]), torch.tensor(bad[1])
        fg.ii_inac, fg.jj_inac = torch.tensor(inac[0]), torch.tensor(inac[1])
        got = {}
        fg.add_factors = lambda ii, jj, remove=False: got.update(ii=ii.clone(), jj=jj.clone(), remove=remove)
        fg.add_proximity_factors(beta=0.3, remove=bool(n), **kw)
        out["ii_%d" % n], out["jj_%d" % n], out["remove_%d" % n] = got["ii"].numpy(), got["jj"].numpy(), np.int64(got["remove"])
        out["args_%d" % n] = np.array([kw["t0"], kw["t1"], kw["rad"], kw["nms"], kw["thresh"]])
    np.savez_compressed(os.path.join(HERE, "proximity_factors.npz"), **out)
    print("proximity_factors: %s edges" % [int(out["ii_%d" % n].shape[0]) for n in range(3)])


def bookkeeping_video(F=8, ht=2, wd=3):
    """mock DepthVideo for the edge-bookkeeping fixture; every buffer row f carries the value f"""
    class Video:
        pass
    v = Video()
    v.ht, v.wd = ht * 8, wd * 8
    col = torch.arange(F).float()
    v.poses = col[:, None].repeat(1, 7).clone()
    v.disps = col[:, None, None].repeat(1, ht, wd).clone()
    v.intrinsics = col[:, None].repeat(1, 4).clone()
    v.nets = col[:, None, None, None].repeat(1, 128, ht, wd).clone()
    v.inps = v.nets.clone() + 100
    v.fmaps = v.nets.clone() + 200
    v.segms = torch.arange(F).int()[:, None, None, None].repeat(1, 1, ht, wd).clone()
    v.segm_filter, v.thresh = True, 0.5

    def reproject(ii, jj):
        val = (torch.as_tensor(ii).float() * 10 + torch.as_tensor(jj).float())[None, :, None, None, None]
        return val.repeat(1, 1, ht, wd, 2), torch.ones(1, len(ii), ht, wd, 1)
    v.reproject = reproject
    import contextlib
    v.get_lock = contextlib.nullcontext                # the reference wraps rm_keyframe's buffer moves in the video lock
    return v


def bookkeeping_script(fg, set_age):
    """the sequence of edge operations both implementations run; returns snapshots after every step"""
    snaps = []

    def snap():
        snaps.append(dict(ii=fg.ii.clone(), jj=fg.jj.clone(), age=fg.age.clone(), ii_inac=fg.ii_inac.clone(), jj_inac=fg.jj_inac.clone(),
                          target=fg.target_cam[0, :, 0, 0, 0].clone(), target_inac=fg.target_cam_inac[0, :, 0, 0, 0].clone(),
                          net=fg.net[0, :, 0, 0, 0].float().clone(), segm=fg.segm[0, :, 0, 0, 0].clone(),
                          weight=fg.weight[0, :, 0, 0, 0].clone()))
    fg.add_neighborhood_factors(0, 5, r=2); snap()
    set_age(fg, [3, 11, 7, 0, 13, 5, 9, 1, 12, 6, 2, 10, 4, 8])
    fg.weight = fg.weight + torch.arange(fg.ii.shape[0]).float()[None, :, None, None, None]      # make per-edge state distinguishable
    fg.corr = object()                                                 # a volume "exists": eviction is armed
    fg.add_factors([5, 4, 5, 0, 5, 6], [4, 5, 3, 1, 4, 5], remove=True); snap()      # two duplicates, 4 new edges -> eviction + store
    fg.rm_keyframe(2); snap()
    fg.rm_factors(torch.tensor([k % 3 == 0 for k in range(fg.ii.shape[0])]), store=False); snap()
    fg.add_factors(torch.tensor([1, 3]), torch.tensor([3, 1])); snap()
    return snaps


def gen_bookkeeping():
    """FactorGraph edge bookkeeping (factor_graph.py:65-225): duplicate filtering, age-based eviction with storage of the
    evicted edges, rm_keyframe's index shifts and buffer moves, rm_factors, on a mock video."""
    import factor_graph as ref_fg
    v = bookkeeping_video()
    fg = ref_fg.FactorGraph(v, None, device="cpu", corr_impl="alt", max_factors=10)

    def set_age(f, ages):
        f.age = torch.tensor(ages)
    snaps = bookkeeping_script(fg, set_age)
    out = {}
    for n, sn in enumerate(snaps):
        for k, t in sn.items():
            out["%s_%d" % (k, n)] = t.numpy()
    out.update(poses=v.poses.numpy(), disps=v.disps.numpy(), nets=v.nets[:, 0, 0, 0].numpy(), fmaps=v.fmaps[:, 0, 0, 0].numpy(),
               segms=v.segms[:, 0, 0, 0].numpy(), n_snaps=np.int64(len(snaps)))
    np.savez_compressed(os.path.join(HERE, "edge_bookkeeping.npz"), **out)
    print("edge_bookkeeping: edges per step %s" % [int(sn["ii"].shape[0]) for sn in snaps])


def filler_case(seed=9):
    """keyframe time stamps / poses and the non-keyframe stamps to fill (shared by generator and test)"""
    from pvo_amd.geom.se3 import SE3
    g = torch.Generator().manual_seed(seed)
    ts = torch.tensor([0.0, 3.0, 4.0, 8.0, 9.0])
    xi = torch.randn(5, 6, generator=g) * torch.tensor([0.3, 0.3, 0.3, 0.05, 0.05, 0.05])
    poses = SE3.exp(torch.cumsum(xi, 0)).data
    stamps = [0, 1, 2, 3, 5, 6.5, 8, 9, 11]                       # on keyframes, between them, after the last one
    return ts, poses, stamps


def gen_filler():
    """PoseTrajectoryFiller.__fill (trajectory_filler.py:35-77) with a recording stand-in for FactorGraph and a zero
    feature encoder: pins the keyframe bracketing, the constant-twist interpolation on se(3), the edges that connect each
    frame to its bracketing keyframes and the six motion-only updates over [N, N+M)."""
    import trajectory_filler as ref_tf
    from pvo_amd.geom.se3 import SE3
    ts, poses, stamps = filler_case()
    N, M, ht, wd = ts.shape[0], len(stamps), 16, 24
    rec = {"calls": []}

    class Counter:
        value = N

    class Video:
        def __setitem__(self, index, item):
            rec["set_index"] = (index.start, index.stop)
            rec["set_tstamp"], rec["set_poses"], rec["set_disp"], rec["set_intr"] = item[0].clone(), item[2].clone(), item[3], item[4].clone()
            self.poses[index] = item[2]
    v = Video()
    v.counter = Counter()
    v.tstamp = torch.cat([ts, torch.zeros(16)])
    v.poses = torch.cat([poses, torch.zeros(16, 7)])

    class Graph:
        def __init__(self, video, update_op, *a, **k):
            pass

        def add_factors(self, ii, jj):
            rec["calls"].append(("add", ii.clone(), jj.clone()))

        def update(self, t0, t1, motion_only=False):
            rec["calls"].append(("update", t0, t1, motion_only, v.counter.value))
    ref_tf.FactorGraph = Graph

    class Net:
        cnet = None
        update = None

        @staticmethod
        def fnet(x):
            return torch.zeros(x.shape[0], x.shape[1], 128, ht // 8, wd // 8)
    filler = ref_tf.PoseTrajectoryFiller(Net, v, device="cpu")
    images = [torch.zeros(3, ht, wd) for _ in stamps]
    intr = [torch.tensor([10.0, 10.0, 12.0, 8.0]) for _ in stamps]
    out = filler._PoseTrajectoryFiller__fill(stamps, images, intr)
    adds = [c for c in rec["calls"] if c[0] == "add"]
    ups = [c for c in rec["calls"] if c[0] == "update"]
    res = dict(init_poses=rec["set_poses"].numpy(), set_index=np.array(rec["set_index"]), set_tstamp=rec["set_tstamp"].numpy(),
               set_intr=rec["set_intr"].numpy(), set_disp=np.float64(rec["set_disp"]), returned=out[0].data.numpy(),
               add0_ii=adds[0][1].numpy(), add0_jj=adds[0][2].numpy(), add1_ii=adds[1][1].numpy(), add1_jj=adds[1][2].numpy(),
               updates=np.array([[u[1], u[2], int(u[3]), u[4]] for u in ups]), counter_after=np.int64(v.counter.value))
    np.savez_compressed(os.path.join(HERE, "trajectory_filler.npz"), **res)
    print("trajectory_filler: %d frames, %d update calls" % (M, len(ups)))


class RecordingGraph:
    """stand-in for FactorGraph in the frontend fixture: records every call (shared by the generator and the test)"""
    log = None
    script = None

    def __init__(self, video, update_op, device="cpu", max_factors=-1, **kw):
        self.video = video
        RecordingGraph.log.append(("init", max_factors))
        self.corr = None
        self.ii = torch.zeros(0, dtype=torch.long); self.age = torch.zeros(0, dtype=torch.long)
        self._ii_h, self._age_h = [], []

    def _set(self, ii, age):
        self.ii, self.age = torch.tensor(ii), torch.tensor(age)
        self._ii_h, self._age_h = list(ii), list(age)

    def add_neighborhood_factors(self, t0, t1, r=3):
        RecordingGraph.log.append(("neigh", t0, t1, r)); self.corr = object(); self._set([0, 1, 2], [0, 0, 0])

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False):
        RecordingGraph.log.append(("prox", t0, t1, rad, nms, round(float(beta), 6), float(thresh), bool(remove)))
        self._set([max(t0, 0), max(t0, 0) + 1, 30], [3, 26, 1])

    def rm_factors(self, mask, store=False):
        m = [bool(x) for x in (mask.tolist() if isinstance(mask, torch.Tensor) else mask)]
        RecordingGraph.log.append(("rm", m, bool(store)))

    def rm_keyframe(self, ix):
        RecordingGraph.log.append(("rm_keyframe", int(ix)))

    def update(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False):
        RecordingGraph.log.append(("update", t0, t1, bool(use_inactive)))


def frontend_video(buf=16, ht=4, wd=6):
    class Video:
        pass
    v = Video()
    v.poses = torch.arange(buf).float()[:, None].repeat(1, 7).clone()
    v.disps = torch.arange(buf).float()[:, None, None].repeat(1, ht, wd).clone() + torch.arange(wd).float() * 0.01
    v.dirty = torch.zeros(buf, dtype=torch.bool)
    v.tstamp = torch.arange(buf).float()
    return v


FRONTEND_DISTANCES = [1.0, 5.0, 0.3, 9.0]           # keyframe test outcomes of four successive updates (thresh 2.25)


def gen_frontend():
    """DroidFrontend.__call__ (droid_frontend.py:36-112) over an initialisation and four updates (two of which drop the
    previous keyframe) with a recording stand-in for FactorGraph: pins the call sequence and arguments, the pose / depth
    guesses for the next frame, the counter and dirty bookkeeping."""
    import contextlib
    import droid_frontend as ref_fe
    from argparse import Namespace

    class Counter:
        value = 0

    class Ready:
        value = 0
    v = frontend_video()
    v.counter, v.ready, v.get_lock = Counter(), Ready(), contextlib.nullcontext
    dist = list(FRONTEND_DISTANCES)
    RecordingGraph.log = []
    v.distance = lambda ii, jj, beta=0.3, bidirectional=True: (RecordingGraph.log.append(("dist", list(ii), list(jj), round(float(beta), 6), bidirectional)), torch.tensor([dist.pop(0)]))[1]
    ref_fe.FactorGraph = RecordingGraph
    net = Namespace(update=None)
    args = Namespace(device="cpu", warmup=5, beta=0.6, frontend_nms=1, keyframe_thresh=2.25, frontend_window=25,
                     frontend_thresh=12.0, frontend_radius=2)
    fe = ref_fe.DroidFrontend(net, v, args)
    snaps = []
    for step in range(10):
        if v.counter.value < 5 or fe.is_initialized:
            v.counter.value += 1                                  # the motion filter appended a keyframe
        fe()
        snaps.append((v.counter.value, fe.t1, int(fe.is_initialized), v.poses[:, 0].clone(), v.disps[:, 0, 0].clone(), v.dirty.clone()))
        if not dist and fe.is_initialized and step > 6:
            break
    out = dict(log=np.array([repr(x) for x in RecordingGraph.log]), counter=np.array([s_[0] for s_ in snaps]), t1=np.array([s_[1] for s_ in snaps]),
               init=np.array([s_[2] for s_ in snaps]), poses=torch.stack([s_[3] for s_ in snaps]).numpy(),
               disps=torch.stack([s_[4] for s_ in snaps]).numpy(), dirty=torch.stack([s_[5] for s_ in snaps]).numpy())
    np.savez_compressed(os.path.join(HERE, "frontend_calls.npz"), **out)
    print("frontend_calls: %d graph calls over %d frontend invocations" % (len(RecordingGraph.log), len(snaps)))


MOTION_SCRIPT = [3.0, 0.5, 1.0, 2.0, 0.2]          # mean flow magnitude the (mock) update operator reports per frame after the first


class MotionNet:
    """mock network for the motion-filter fixture: deterministic features, an update operator with scripted flow"""

    def __init__(self, ht, wd):
        self.h, self.w = ht // 8, wd // 8
        self.calls = []
        self.script = list(MOTION_SCRIPT)

    def fnet(self, x):
        return x.mean(dim=2, keepdim=True).mean(dim=(3, 4), keepdim=True).expand(1, x.shape[1], 128, self.h, self.w) + 0.0

    def cnet(self, x):
        return x.mean(dim=2, keepdim=True).mean(dim=(3, 4), keepdim=True).expand(1, x.shape[1], 256, self.h, self.w) * 2.0

    def update(self, net, inp, corr, **kw):
        self.calls.append((tuple(net.shape), tuple(inp.shape), float(net.float().mean()), float(inp.float().mean())))
        mag = self.script.pop(0)
        delta = torch.zeros(1, 1, self.h, self.w, 4)
        delta[..., 0] = mag                                      # |delta[..., 0:2]| = mag everywhere
        return net, delta, torch.zeros(1, 1, self.h, self.w, 2), torch.zeros(1, 1, self.h, self.w, 2)


def motion_frames(n=6, ht=32, wd=48):
    g = torch.Generator().manual_seed(10)
    return [(float(t), torch.randint(0, 256, (3, ht, wd), generator=g).int(), torch.tensor([40.0, 40.0, 24.0, 16.0]) * (1 + 0.01 * t), None)
            for t in range(n)]


def gen_motion_filter():
    """MotionFilter.track (motion_filter.py:46-87) with a mock network: which frames become keyframes (mean one-step flow
    above `thresh`), and what is appended to the video for them (identity pose and unit depth only for the first)."""
    import motion_filter as ref_mf
    ht, wd = 32, 48
    appended = []

    class Counter:
        value = 0

    class Video:
        counter = Counter()

        def append(self, *item):
            appended.append(item); Video.counter.value += 1

    class FakeCorr:
        def __init__(self, f1, f2, *a, **k):
            pass

        def __call__(self, coords):
            return torch.zeros(1, 1, 196, ht // 8, wd // 8)
    ref_mf.CorrBlock = FakeCorr
    net = MotionNet(ht, wd)
    mf = ref_mf.MotionFilter(net, Video(), thresh=1.75, device="cpu")
    counts = []
    for t, image, intr, segm in motion_frames():
        mf.track(t, image, None, intr, segm)
        counts.append(mf.count)
    out = dict(n_appended=np.int64(len(appended)), counts=np.array(counts), tstamps=np.array([a[0] for a in appended]),
               has_pose=np.array([a[2] is not None for a in appended]), has_disp=np.array([a[3] is not None for a in appended]),
               intr=torch.stack([a[4] for a in appended]).numpy(), fmap_mean=np.array([float(a[5].float().mean()) for a in appended]),
               net_mean=np.array([float(a[6].float().mean()) for a in appended]), inp_mean=np.array([float(a[7].float().mean()) for a in appended]),
               op_calls=np.array([[c[2], c[3]] for c in net.calls]))
    np.savez_compressed(os.path.join(HERE, "motion_filter.npz"), **out)
    print("motion_filter: %d of %d frames became keyframes" % (len(appended), len(counts)))


def video_script(v, set_counter, log):
    """the DepthVideo calls both implementations answer (native calls are recorded by `log`)"""
    g = torch.Generator().manual_seed(21)
    F = 6
    v.poses[:F] = torch.randn(F, 7, generator=g)
    v.disps[:F] = torch.rand(F, v.disps.shape[1], v.disps.shape[2], generator=g) + 0.5
    v.intrinsics[:F] = torch.tensor([20.0, 21.0, 12.0, 8.0])
    set_counter(v, F)
    res = {}
    res["d_pairs"] = v.distance([0, 2, 5], [1, 4, 3], beta=0.4, bidirectional=True)
    res["d_one_way"] = v.distance(torch.tensor([1, 3]), torch.tensor([0, 2]), beta=0.7, bidirectional=False)
    res["d_matrix"] = v.distance(beta=0.3)
    ii, jj = torch.tensor([0, 1, 3, 4]), torch.tensor([1, 0, 4, 2])
    ht, wd = v.disps.shape[1:]
    tg, wt = torch.randn(4, 2, ht, wd, generator=g), torch.rand(4, 2, ht, wd, generator=g)
    eta = torch.rand(4, ht, wd, generator=g)
    v.ba(tg, wt, eta, ii, jj, t0=1, t1=5, itrs=3, lm=1e-3, ep=0.2, motion_only=False)
    v.ba(tg, wt, None, ii, jj)                                       # defaults: t1 from the edges, unit-row eta of 1e-7
    v.normalize()
    res["poses"], res["disps"], res["dirty"] = v.poses[:F].clone(), v.disps[:F].clone(), v.dirty[:F + 1].clone()
    return res


def make_native_recorder(log):
    def frame_distance(poses, disps, intr, ii, jj, beta):
        log.append(("frame_distance", tuple(poses.shape), tuple(intr.tolist()), ii.tolist(), jj.tolist(), round(float(beta), 6)))
        return (ii * 10 + jj).float() + float(poses[0, 0])

    def ba(poses, disps, intr, target, weight, eta, ii, jj, t0, t1, itrs, lm, ep, motion_only, *a, **k):
        log.append(("ba", tuple(intr.tolist()), tuple(eta.shape), float(eta.flatten()[0]), ii.tolist(), jj.tolist(), int(t0), int(t1), int(itrs),
                    round(float(lm), 9), round(float(ep), 9), bool(motion_only), float(target.sum()), float(weight.sum())))
        disps[1, 0, :3] = torch.tensor([-1.0, 0.0005, 0.3])           # the caller clamps at 0.001
        return [None, None]
    return frame_distance, ba


def gen_depth_video():
    """DepthVideo.distance / ba / normalize (depth_video.py:145-214) with recording stand-ins for the two native calls:
    pins argument order and defaults (shared intrinsics[0], bidirectional averaging, t1 and eta defaults, the 0.001 clamp)."""
    import depth_video as ref_dv
    log = []
    fd, ba = make_native_recorder(log)
    ref_dv.droid_backends.frame_distance, ref_dv.droid_backends.ba = fd, ba
    v = ref_dv.DepthVideo(image_size=[32, 48], buffer=8, device="cpu")

    def set_counter(video, n):
        video.counter.value = n
    res = video_script(v, set_counter, log)
    out = {k: t.numpy() for k, t in res.items()}
    out["log"] = np.array([repr(x) for x in log])
    np.savez_compressed(os.path.join(HERE, "depth_video_calls.npz"), **out)
    print("depth_video_calls: %d native calls" % len(log))


def losses_case(seed=21):
    """inputs of the training-loss fixture, regenerated from the seed by generator and test alike: a 4-frame clip at
    48x64 (6x8 maps), 3 unrolled steps, the |i-j| <= 2 graph"""
    from collections import OrderedDict
    from pvo_amd.geom.se3 import SE3
    g = torch.Generator().manual_seed(seed)
    N, H, W, n = 4, 48, 64, 3
    h, w = H // 8, W // 8
    graph = OrderedDict((i, [j for j in range(N) if i != j and abs(i - j) <= 2]) for i in range(N))
    E = sum(len(v) for v in graph.values())
    xi = torch.tensor([0.05, 0.01, 0.02, 0.004, 0.01, -0.006])
    Ps = SE3(torch.stack([SE3.exp(k * xi).data for k in range(N)])[None])
    low = torch.rand(1, N, 6, 8, generator=g) * 0.6 + 0.4
    disps = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True)
    intr = torch.tensor([50.0, 50.0, W / 2.0, H / 2.0]).view(1, 1, 4).repeat(1, N, 1)
    images = torch.nn.functional.interpolate(torch.rand(N, 3, 12, 16, generator=g) * 255, size=(H, W), mode="bilinear")[None]
    poses_est = [SE3(Ps.data + 0.02 * (n - k) * torch.randn(1, N, 7, generator=g) * torch.tensor([1, 1, 1, .2, .2, .2, 0.0])) for k in range(n)]
    for G in poses_est:
        G.data[..., 3:] = G.data[..., 3:] / G.data[..., 3:].norm(dim=-1, keepdim=True)
    disps_est = [disps * (1 + 0.05 * (n - k) * torch.randn(1, N, H, W, generator=g)).clamp(0.5, 1.5) for k in range(n)]
    residuals = [torch.randn(1, E, h, w, 2, generator=g) * (n - k) for k in range(n)]
    full_flows = [torch.randn(1, E, h, w, 2, generator=g) * 1.5 for _ in range(n)]
    masks = [torch.sigmoid(torch.randn(1, E, H, W, 2, generator=g)) for _ in range(n)]
    gt_masks = (torch.rand(1, N, H, W, 1, generator=g) > 0.3).float()
    gt_vals = (torch.rand(1, N, H, W, 1, generator=g) > 0.1).float()
    fo = torch.cat([torch.randn(1, E // 2, h, w, 2, generator=g), (torch.rand(1, E // 2, h, w, 1, generator=g) > 0.2).float()], -1)
    bo = torch.cat([torch.randn(1, E // 2, h, w, 2, generator=g), (torch.rand(1, E // 2, h, w, 1, generator=g) > 0.2).float()], -1)
    aff = [torch.cat([1 + 0.1 * torch.randn(1, E, 1, generator=g), 0.5 + 0.05 * torch.randn(1, E, 1, generator=g)], -1) for _ in range(n)]
    return dict(graph=graph, Ps=Ps, disps=disps, intr=intr, images=images, poses_est=poses_est, disps_est=disps_est,
                residuals=residuals, full_flows=full_flows, masks=masks, gt_masks=gt_masks, gt_vals=gt_vals, fo=fo, bo=bo, aff=aff, N=N)


def gen_losses():
    """the reference's training losses (geom/losses.py) on losses_case(): every loss train.py can combine, with its
    metrics.  Fixture-time stand-ins for the two lietorch groups the METRICS of geodesic_loss touch (losses.py:12-22,71):
    Sim3(X) keeps X's translation / rotation and a unit scale, SO3(q).log() is the rotation part of SE3's log."""
    import lietorch
    from pvo_amd.geom.se3 import SE3

    class Sim3:
        def __init__(self, X):
            self.data = torch.cat([X.data, torch.ones_like(X.data[..., :1])], -1)
        def detach(self):
            return self

    class SO3:
        def __init__(self, q):
            self.q = q
        def log(self):
            return SE3(torch.cat([torch.zeros_like(self.q[..., :3]), self.q], -1)).log()[..., 3:]
    lietorch.Sim3, lietorch.SO3 = Sim3, SO3
    import importlib
    import geom.losses as L
    L = importlib.reload(L)
    L.Sim3, L.SO3 = Sim3, SO3
    c = losses_case()
    out = {}

    def put(name, res):
        loss, metrics = res
        out[name] = np.float64(float(loss))
        for k, v in metrics.items():
            out[name + "/" + k] = np.float64(v)
    ssim = L.SSIM()
    put("residual", L.residual_loss(c["residuals"]))
    put("geodesic", L.geodesic_loss(c["Ps"], c["poses_est"], c["graph"], do_scale=False))
    put("cam_flow", L.cam_flow_loss(c["Ps"], c["disps"], c["poses_est"], c["disps_est"], c["intr"], c["graph"]))
    put("flow", L.flow_loss(c["fo"], c["bo"], c["full_flows"], c["graph"]))
    put("photo_sup_ds", L.photo_loss(c["images"], c["full_flows"], c["gt_vals"], c["graph"], "semisup", ssim=None, aff_params=None, downsample=True))
    put("photo_aff_ssim", L.photo_loss(c["images"], c["full_flows"], c["gt_vals"], c["graph"], "sup", ssim=ssim, aff_params=c["aff"], downsample=True, mean_mask=True))
    put("photo_cam", L.photo_loss_cam(c["images"], c["poses_est"], c["disps_est"], c["intr"], c["graph"], "semisup", c["gt_masks"], ssim=ssim))
    put("gt_label", L.gt_label_loss(c["gt_masks"], c["gt_vals"], c["masks"], c["graph"]))
    put("gt_label_mean_mask", L.gt_label_loss(c["gt_masks"], c["gt_vals"], c["masks"], c["graph"], mean_mask=True))
    put("ce_reg", L.ce_reg_loss(c["masks"]))
    put("consistency", L.consistency_loss(c["masks"], c["N"], c["graph"]))
    # the unsupervised mode's label / occlusion machinery (CPU tensors: the reference moves them there itself)
    art = L.unsup_art_label(c["poses_est"], c["disps_est"], c["intr"].clone(), c["full_flows"], c["graph"], downsample=True)
    for k, a in enumerate(art):
        out["art_label_%d" % k] = a.numpy()
    masks_cpu = c["masks"]
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                  # (losses.py:445,447 call .cuda() on the labels)
    try:
        put("art_label", L.art_label_loss(art, masks_cpu, downsample=True))
    finally:
        torch.Tensor.cuda = real_cuda
    for tag in ("ph_loss", "cam_ph_loss"):
        ds = tag == "ph_loss"
        vals = L.unsup_occ_vals(c["poses_est"], c["disps_est"], c["intr"].clone(), ds, c["graph"] if ds else None, tag)
        for k, v in enumerate(vals):
            out["occ_%s_%d" % (tag, k)] = v.numpy()
        if ds:
            dy = L.unsup_dy_vals(vals, c["gt_masks"][..., 0], c["graph"])
            for k, v in enumerate(dy):
                out["dy_%d" % k] = v.numpy()
    x, y = c["images"][0, :2] / 255.0, c["images"][0, 1:3] / 255.0
    out["ssim_map"] = ssim(x, y).numpy()
    np.savez_compressed(os.path.join(HERE, "train_losses.npz"), **out)
    print("train_losses.npz:", len(out), "entries")


def frame_graph_case(seed=31, N=7, H=64, W=96):
    """a clip for the training-time frame graph (shared by generator and test): poses drift unevenly so that some
    non-neighbours are co-visible and others are not"""
    from pvo_amd.geom.se3 import SE3
    g = torch.Generator().manual_seed(seed)
    xi = torch.tensor([0.09, 0.01, 0.03, 0.004, 0.02, -0.006])
    steps = torch.cumsum(torch.rand(N, generator=g) * 1.5, 0)
    poses = torch.stack([SE3.exp(float(t) * xi).data for t in steps])[None]
    low = torch.rand(1, N, 4, 6, generator=g) * 0.6 + 0.3
    disps = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True)
    intr = torch.tensor([80.0, 80.0, W / 2.0, H / 2.0]).view(1, 1, 4).repeat(1, N, 1)
    return poses, disps, intr


def gen_frame_graph():
    """geom/graph_utils.py build_frame_graph + data_readers/rgbd_utils.py compute_distance_matrix_flow (the reference
    calls .cuda() on its inputs: a no-op at fixture time)"""
    import geom.graph_utils as gu
    from data_readers import rgbd_utils as ru
    poses, disps, intr = frame_graph_case()
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        out = {}
        for need_inv in (False, True):
            d = ru.compute_distance_matrix_flow(poses[0].numpy(), disps[0][:, 3::8, 3::8].numpy(), (intr[0] / 8.0).numpy(), need_inv)
            out["dist_inv%d" % need_inv] = d
            for num, thresh in ((20, 24.0), (40, 24.0), (30, 6.0)):
                g = gu.build_frame_graph(poses, disps, intr, num=num, thresh=thresh, need_inv=need_inv)
                ii, jj, _ = gu.graph_to_edge_list(g)
                out["edges_inv%d_%d_%g" % (need_inv, num, thresh)] = torch.stack([ii, jj]).numpy()
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(HERE, "frame_graph.npz"), **out)
    print("frame_graph.npz:", {k: v.shape for k, v in out.items()})


def gen_frame_distance():
    """frame_distance's two terms (droid_kernels.cu:497-636) from the reference's Python geometry: the reprojection term is
    the mean norm of pops.induced_flow; the TRANSLATION-ONLY term projects X + d t_ij, which is what pops.projective_transform
    computes for a relative pose with identity rotation - evaluated here edge by edge on the two-frame pose set
    [identity, (t_ij, identity)] with t_ij the translation of G_j G_i^-1 (lietorch operators -> pvo_amd.geom.se3, pinned)."""
    from pvo_amd.geom.se3 import SE3
    import geom.projective_ops as pops
    out = {}
    for name, seed, P, ht, wd in (("a", 40, 5, 8, 10), ("b", 41, 6, 6, 9)):
        s = make_scene(seed, P, ht, wd)
        G = SE3(s["poses"][None])
        intr_all = s["intr"][None, None].repeat(1, P, 1)
        flow, _ = pops.induced_flow(G, s["disps"][None], intr_all, s["ii"], s["jj"])
        full = flow[0].norm(dim=-1).mean(dim=(1, 2))
        Gij = G[:, s["jj"]] * G[:, s["ii"]].inv()
        tonly = []
        for e in range(s["ii"].shape[0]):
            two = torch.zeros(1, 2, 7); two[..., 6] = 1.0
            two[0, 1, :3] = Gij.data[0, e, :3]
            d2 = torch.stack([s["disps"][s["ii"][e]], s["disps"][s["ii"][e]]])[None]
            f, v = pops.induced_flow(SE3(two), d2, intr_all[:, :2], torch.tensor([0]), torch.tensor([1]))
            assert bool((v == 1).all())
            tonly.append(f[0, 0].norm(dim=-1).mean())
        for k in ("poses", "disps", "intr", "ii", "jj"):
            out[name + "_" + k] = s[k].numpy()
        out[name + "_full_mean"] = full.numpy()
        out[name + "_tonly_mean"] = torch.stack(tonly).numpy()
    np.savez_compressed(os.path.join(HERE, "frame_distance_terms.npz"), **out)
    print("frame_distance_terms.npz:", {k: v.shape for k, v in out.items() if k.endswith("mean")})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be generated in the build container")
    install_substitutes()
    torch.set_num_threads(4)
    gen_ba()
    gen_corr()
    gen_update_op()
    gen_graph()
    gen_factor_graph_glue()
    gen_droidnet()
    gen_lowmem_glue()
    gen_proximity()
    gen_bookkeeping()
    gen_filler()
    gen_frontend()
    gen_motion_filter()
    gen_depth_video()
    gen_losses()
    gen_frame_graph()
    gen_frame_distance()
