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
