"""Train the VO network — the role of the reference's VO_Module/train.py, same arguments and step structure:

    python tools/train.py --gpus 0,1,2,3 --steps 20000            # one process per GPU (spawned), DDP over RCCL
    python -m torch.distributed.run --nproc-per-node 4 tools/train.py ...   # or under a launcher (RANK / WORLD_SIZE in the env)

Per step (train.py:108-262): a clip of `n_frames`, poses initialised to frame 1, unit depth, a frame graph (co-visibility
graph or |i-j| <= 2, train.py:128-140), random restarts (`restart_prob`), `DroidNet.forward` for `iters` unrolled updates
(HIP correlation lookup forward / backward, PyTorch BA), the mode's losses (geom/losses.py), gradient clipping, Adam +
OneCycleLR.  `--corr_dtype bfloat16` keeps the all-pairs volume in bf16 (BASELINE.json configs[4]).

Data: the VKITTI2 / TartanAir readers are outside this build's scope and no dataset exists in the
environment, so clips come from `pvo_amd.synthetic.TrainClips` in the readers' item layout; a `--datapath` other than
"synthetic" raises.  Checkpoints are written as the reference writes them (`checkpoints/<name>_<step>.pth`, DDP-wrapped
state-dict keys) and load into the reference's DroidNet unchanged (state-dict compatible, tests/test_droidnet.py).
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def str2bool(v):
    """the reference declares its switches with argparse `type=bool` (train.py:317-393): any non-empty string, 'False' included, is
    True there, so its defaults can only be switched ON from the command line.  Same option names and defaults here, but the value
    is parsed: --ssim False turns SSIM off."""
    if isinstance(v, bool):
        return v
    if v.lower() in ("1", "true", "t", "yes", "y", "on"):
        return True
    if v.lower() in ("0", "false", "f", "no", "n", "off", ""):
        return False
    raise argparse.ArgumentTypeError("expected a boolean, got %r" % v)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    # (defaults are train.py:279-343's)
    p.add_argument("--name", default="vkitti2_dy_train")
    p.add_argument("--datapath", default="synthetic")
    p.add_argument("--need_inv", type=str2bool, default=False)
    p.add_argument("--gpus", type=str, default="0")
    p.add_argument("--mode", type=str, default="semisup", help="sup, semisup, unsup")
    p.add_argument("--lr", type=float, default=0.0005)
    p.add_argument("--steps", type=int, default=20000)
    p.add_argument("--occ_ph", type=str2bool, default=False)
    p.add_argument("--ckpt")
    p.add_argument("--flow_label", type=str2bool, default=False)
    p.add_argument("--aug_graph", type=str2bool, default=True)
    p.add_argument("--use_aff_bri", type=str2bool, default=False)
    p.add_argument("--downsample", type=str2bool, default=True)
    p.add_argument("--ssim", type=str2bool, default=True)
    p.add_argument("--ce_reg", type=str2bool, default=False)
    p.add_argument("--con_loss", type=str2bool, default=False)
    p.add_argument("--ph_loss", type=str2bool, default=True)
    p.add_argument("--batch", type=int, default=1)
    p.add_argument("--iters", type=int, default=15)
    p.add_argument("--clip", type=float, default=2.5)
    p.add_argument("--n_frames", type=int, default=6)
    for name, val in (("w1", 40.0), ("w2", 0.01), ("w3", 0.20), ("w4", 5.0), ("w5", 0.001), ("w6", 0.05), ("w7", 0.01),
                      ("w8", 0.05), ("w9", 0.01), ("w10", 100.0)):
        p.add_argument("--" + name, type=float, default=val)
    p.add_argument("--fmin", type=float, default=8.0)
    p.add_argument("--fmax", type=float, default=96.0)
    p.add_argument("--edges", type=int, default=20)
    p.add_argument("--restart_prob", type=float, default=0.2)
    # this build
    p.add_argument("--crop_size", type=int, nargs=2, default=[200, 400])
    p.add_argument("--corr_dtype", default="bfloat16", choices=["float32", "bfloat16", "float16"])
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu: gloo, for the plumbing tests")
    p.add_argument("--port", type=int, default=12356)
    p.add_argument("--dist_backend", default="auto", choices=["auto", "nccl", "gloo"],
                   help="auto: nccl (RCCL) on the GPU, gloo on the CPU; gloo with --device cuda lets several ranks share one GPU (tests)")
    p.add_argument("--save_every", type=int, default=2000)
    p.add_argument("--log_every", type=int, default=100)
    p.add_argument("--out_dir", default="checkpoints")
    args = p.parse_args(argv)
    args.world_size = len(args.gpus.split(","))
    return args


def make_graph(args, poses, disps, intrinsics, rng_state):
    """train.py:128-140"""
    from pvo_amd.geom.graph_utils import build_frame_graph
    N = args.n_frames
    if args.aug_graph:
        if rng_state.random() < 0.5:
            return build_frame_graph(poses, disps, intrinsics, num=args.edges, need_inv=args.need_inv)
        r = 2
    else:
        r = 1
    return OrderedDict((i, [j for j in range(N) if i != j and abs(i - j) <= r]) for i in range(N))


def objective(args, L, out, batch, graph, ssim, step):
    """the weighted sum of train.py:142-236 for one forward pass; returns (loss, metrics)"""
    images, Ps, disps, intrinsics, gt_masks, gt_vals = batch
    poses_est, disps_est, residuals = out[0], out[1], out[2]
    full_flows = out[3] if (args.flow_label or args.ph_loss) else None
    masks = out[4] if full_flows is not None else out[3]
    aff = out[5] if (args.use_aff_bri and full_flows is not None) else None
    metrics, loss = {}, 0.0
    res_loss, m = L.residual_loss(residuals); metrics.update(m)
    loss = loss + args.w2 * res_loss
    # (`late` is keyed on the GLOBAL step here; the reference keys it on i_batch, the index inside the current pass over its data
    # loader, train.py:196 - with synthetic clips there is no epoch boundary to restart it at)
    late = args.occ_ph and step > args.steps * 0.75
    if args.mode == "sup":
        geo, m = L.geodesic_loss(Ps, poses_est, graph, do_scale=False); metrics.update(m)
        cam, m = L.cam_flow_loss(Ps, disps, poses_est, disps_est, intrinsics, graph); metrics.update(m)
        lab, m = L.gt_label_loss(gt_masks, gt_vals, masks, graph); metrics.update(m)
        loss = loss + args.w1 * geo + args.w3 * cam + args.w9 * lab
        if args.ph_loss:
            ph, m = L.photo_loss(images, full_flows, gt_vals, graph, "sup", ssim=None, aff_params=aff, downsample=args.downsample)
            metrics.update(m); loss = loss + args.w4 * ph
    elif args.mode == "semisup":
        cam_ph, m = L.photo_loss_cam(images, poses_est, disps_est, intrinsics, graph, "semisup", gt_masks, ssim=ssim); metrics.update(m)
        lab, m = L.gt_label_loss(gt_masks, gt_vals, masks, graph); metrics.update(m)
        loss = loss + args.w10 * cam_ph + args.w9 * lab
        if args.ph_loss:
            if late:
                vals = L.unsup_dy_vals(L.unsup_occ_vals(poses_est, disps_est, intrinsics, args.downsample, graph, "ph_loss"), gt_masks, graph)
                ph, m = L.photo_loss(images, full_flows, vals, graph, "unsup", ssim=None, aff_params=aff, downsample=args.downsample)
            else:
                ph, m = L.photo_loss(images, full_flows, gt_vals, graph, "semisup", ssim=None, aff_params=aff, downsample=args.downsample)
            metrics.update(m); loss = loss + args.w4 * ph
    elif args.mode == "unsup":
        art = L.unsup_art_label(poses_est, disps_est, intrinsics, full_flows, graph, downsample=args.downsample)
        use_one = not late
        ph_vals = L.unsup_occ_vals(poses_est, disps_est, intrinsics, args.downsample, graph, "ph_loss", use_one=use_one)
        cam_vals = L.unsup_occ_vals(poses_est, disps_est, intrinsics, False, None, "cam_ph_loss", use_one=use_one)
        cam_ph, m = L.photo_loss_cam(images, poses_est, disps_est, intrinsics, graph, "unsup", cam_vals, ssim=ssim); metrics.update(m)
        al, m = L.art_label_loss(art, masks, downsample=args.downsample); metrics.update(m)
        loss = loss + args.w10 * cam_ph + args.w6 * al
        if args.ph_loss:
            if not use_one:
                ph_vals = L.unsup_dy_vals(ph_vals, art, graph)
            ph, m = L.photo_loss(images, full_flows, ph_vals, graph, "unsup", ssim=None, aff_params=aff, downsample=args.downsample)
            metrics.update(m); loss = loss + args.w4 * ph
    else:
        raise ValueError("unknown mode %r" % args.mode)
    if args.ce_reg:
        ce, m = L.ce_reg_loss(masks); metrics.update(m); loss = loss + args.w5 * ce
    if args.con_loss:
        con, m = L.consistency_loss(masks, args.n_frames, graph); metrics.update(m); loss = loss + args.w7 * con
    return loss, metrics


def train(rank, args, report=None):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from pvo_amd.droid_net import DroidNet
    from pvo_amd.geom import losses as L
    from pvo_amd.geom.se3 import SE3
    from pvo_amd.logger import Logger
    from pvo_amd.synthetic import TrainClips

    world = args.world_size
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if launched:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(args.port)
        local = int(args.gpus.split(",")[rank])
    cuda = args.device == "cuda"
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(dev)
    backend = ("nccl" if cuda else "gloo") if args.dist_backend == "auto" else args.dist_backend
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    torch.manual_seed(0)                                             # train.py:40: every rank starts from the same weights
    rng = np.random.default_rng(12345)                               # restarts (train.py:52)
    graph_rng = np.random.default_rng(777 + rank)
    try:
        model = DroidNet(args.use_aff_bri).to(dev).train()
        ddp = DDP(model, device_ids=[local] if cuda else None, find_unused_parameters=False)
        if args.ckpt:
            ddp.load_state_dict(torch.load(args.ckpt, map_location=dev))
        ssim = L.SSIM().to(dev) if args.ssim else None
        if args.datapath != "synthetic":
            raise NotImplementedError("dataset readers are not part of this build (SURVEY.md section 2); use --datapath synthetic")
        db = TrainClips(args.n_frames, tuple(args.crop_size), length=max(64, args.batch * world * 8), seed=0)
        sampler = DistributedSampler(db, shuffle=True, num_replicas=world, rank=rank)
        loader = DataLoader(db, batch_size=args.batch, sampler=sampler, num_workers=0)
        opt = torch.optim.Adam(ddp.parameters(), lr=args.lr, weight_decay=1e-5)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, args.lr, args.steps, pct_start=0.01, cycle_momentum=False)
        logger = Logger(args.name, sched, sum_freq=args.log_every)
        corr_dtype = {"float32": None, "bfloat16": torch.bfloat16, "float16": torch.float16}[args.corr_dtype] if cuda else None
        total, epoch, step_ms = 0, 0, []
        while total < args.steps:
            sampler.set_epoch(epoch); epoch += 1
            for item in loader:
                t_start = time.perf_counter()
                opt.zero_grad()
                images, poses, disps, intrinsics, gt_masks, gt_vals, segments = [x.to(dev) for x in item]
                Ps = SE3(poses).inv() if args.need_inv else SE3(poses)
                Gs = SE3.IdentityLike(Ps)
                graph = make_graph(args, poses, disps, intrinsics, graph_rng)
                Gs.data[:, 0] = Ps.data[:, 0].clone()                 # train.py:143-145: first pose fixed, the rest start at pose 1
                Gs.data[:, 1:] = Ps.data[:, [1]].clone()
                disp0 = torch.ones_like(disps[:, :, 3::8, 3::8])
                r, first = 0.0, True
                while first or r < args.restart_prob:                 # random restarts (train.py:148-150; at least one pass,
                    first = False                                     #  also for restart_prob = 0, where the reference's loop never runs)
                    r = rng.random()
                    want_flow = args.flow_label or args.ph_loss
                    out = ddp(Gs, images, disp0, intrinsics / 8.0, graph, num_steps=args.iters, fixedp=2, ret_flow=want_flow,
                              downsample=args.downsample, **({"segments": segments} if want_flow else {}), corr_dtype=corr_dtype)
                    loss, metrics = objective(args, L, out, (images, Ps, disps, intrinsics, gt_masks, gt_vals), graph, ssim, total)
                    loss.backward()
                    Gs = out[0][-1].detach()
                    disp0 = out[1][-1][:, :, 3::8, 3::8].detach()
                torch.nn.utils.clip_grad_norm_(ddp.parameters(), args.clip)
                opt.step(); sched.step()
                total += 1
                if cuda:
                    torch.cuda.synchronize(dev)
                step_ms.append(1e3 * (time.perf_counter() - t_start))
                metrics["loss"] = float(loss)
                if rank == 0:
                    logger.push(metrics)
                    if total % args.save_every == 0:
                        os.makedirs(args.out_dir, exist_ok=True)
                        torch.save(ddp.state_dict(), os.path.join(args.out_dir, "%s_%06d.pth" % (args.name, total)))
                if total >= args.steps:
                    break
        if rank == 0:
            os.makedirs(args.out_dir, exist_ok=True)
            torch.save(ddp.state_dict(), os.path.join(args.out_dir, "%s_final.pth" % args.name))
            print("trained %d steps on %d rank(s): median %.1f ms/step, last loss %.4f" % (total, world, float(np.median(step_ms)), metrics["loss"]))
        if report is not None:
            report[rank] = dict(steps=total, loss=metrics["loss"], ms=float(np.median(step_ms)),
                                w0=float(next(model.parameters()).detach().double().sum()), history=list(logger.history))
    finally:
        dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    print(args)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        train(0, args)
    elif args.world_size == 1:
        train(0, args)
    else:
        import torch.multiprocessing as mp
        mp.spawn(train, nprocs=args.world_size, args=(args,))


if __name__ == "__main__":
    main()
