"""Frame-graph helpers used by the training loop (VO_Module/droid_slam/geom/graph_utils.py:10-34)."""
import torch


def graph_to_edge_list(graph):
    """OrderedDict {u: [v, ...]} -> (ii, jj, kk): sources, targets and the source's ordinal."""
    ii, jj, kk = [], [], []
    for s, (u, vs) in enumerate(graph.items()):
        ii += [u] * len(vs)
        jj += list(vs)
        kk += [s] * len(vs)
    return torch.as_tensor(ii), torch.as_tensor(jj), torch.as_tensor(kk)


def keyframe_indicies(graph):
    return torch.as_tensor(list(graph.keys()))


def neighbourhood_graph(n, r, device="cpu"):
    """all ordered pairs with 1 <= |i-j| <= r"""
    idx = torch.arange(n, device=device)
    ii, jj = idx[:, None].expand(n, n).reshape(-1), idx[None, :].expand(n, n).reshape(-1)
    d = (ii - jj).abs()
    keep = (d >= 1) & (d <= r)
    return ii[keep], jj[keep]


def compute_distance_matrix_flow(poses, disps, intrinsics, need_inv=True):
    """[N,N] mean magnitude of the camera-induced flow between every ordered frame pair, both directions pooled, clamped
    at 100 px; inf where fewer than 70 % of the pixels are valid (data_readers/rgbd_utils.py:110-152).  poses [N,7],
    disps [N,h,w], intrinsics [N,4] at the resolution of `disps`; stays on the inputs' device."""
    from . import projective_ops as pops
    from .se3 import SE3
    poses = torch.as_tensor(poses).float()
    G = SE3(poses[None])
    if need_inv:
        G = G.inv()
    disps, intr = torch.as_tensor(disps).float()[None].to(poses.device), torch.as_tensor(intrinsics).float()[None].to(poses.device)
    N = poses.shape[0]
    idx = torch.arange(N, device=poses.device)
    ii, jj = idx[:, None].expand(N, N).reshape(-1), idx[None, :].expand(N, N).reshape(-1)
    f1, v1 = pops.induced_flow(G, disps, intr, ii, jj)
    f2, v2 = pops.induced_flow(G, disps, intr, jj, ii)
    mag = torch.stack([f1, f2], dim=2).norm(dim=-1).clamp(max=100.0)[0].reshape(N * N, -1)
    val = torch.stack([v1, v2], dim=2)[0].reshape(N * N, -1)
    d = (mag * val).mean(-1) / val.mean(-1)
    d[val.mean(-1) < 0.7] = float("inf")
    return d.view(N, N)


def build_frame_graph(poses, disps, intrinsics, num=16, thresh=24.0, r=2, need_inv=True):
    """training-time frame graph (geom/graph_utils.py:37-68): every frame is linked to its temporal neighbours within r,
    then the co-visible pairs with the smallest flow distance are added until `num` edges (distance below `thresh`).
    poses [1,N,7], disps [1,N,H,W] and intrinsics [1,N,4] at image resolution, as the data loader yields them."""
    from collections import OrderedDict
    N = poses.shape[1]
    d = compute_distance_matrix_flow(poses[0], disps[0][:, 3::8, 3::8], intrinsics[0] / 8.0, need_inv).cpu()
    graph, count = OrderedDict(), 0
    for i in range(N):
        graph[i] = [j for j in range(i - r, i + r + 1) if 0 <= j < N and j != i]
        count += len(graph[i])
        d[i, i] = float("inf")
        for j in graph[i]:
            d[i, j] = float("inf")
    while count < num:
        ix = int(torch.argmin(d))
        i, j = ix // N, ix % N
        if not d[i, j] < thresh:
            break
        graph[i].append(j)
        d[i, j] = float("inf")
        count += 1
    return graph
