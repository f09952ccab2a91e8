"""The pose solve alone (pvo_ba_finish, motion_only: no back-substitution of depths) on synthetic SPD systems of every window size,
against numpy's fp64 solve: prints the largest relative error of dx per size and solver form.

    python tools/solve_check.py [solver]        # solver: dense (default choice up to 29 poses) | blocked | wave | pipe | twin | blocks (48 x 48 blocks over many workgroups, beyond 21 poses)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvo_amd import droid_backends as db  # noqa: E402

FIX = float(1 << 28)
LAST_US = 0.0


def system(P, seed, coupling=1.0):
    """a dense SPD pose system in the library's fixed-point image: lower block triangle (diagonal blocks whole) + rhs"""
    g = np.random.default_rng(seed)
    n = 6 * P
    M = g.standard_normal((n, n + 8)) * coupling
    A = M @ M.T + n * np.eye(n)
    b = g.standard_normal(n) * 10.0
    Aq = np.rint(A * FIX) / FIX
    bq = np.rint(b * FIX) / FIX
    sysm = np.zeros(n * n + n, dtype=np.int64)
    S = np.rint(Aq * FIX).astype(np.int64)
    for r in range(n):
        for c in range(n):
            if c // 6 > r // 6:
                S[r, c] = 0
    sysm[:n * n] = S.reshape(-1)
    sysm[n * n:] = np.rint(bq * FIX).astype(np.int64)
    return Aq, bq, sysm


def solve(P, seed=0, lm=1e-4, ep=0.1, dev="cuda:0"):
    A, b, sysm = system(P, seed)
    n = 6 * P
    want = np.linalg.solve(A + np.diag(ep + lm * np.diag(A)), b)
    F, ht, wd = P + 1, 4, 4
    poses = torch.zeros(F, 7, device=dev); poses[:, 6] = 1.0
    disps = torch.ones(F, ht, wd, device=dev)
    ii = torch.zeros(1, dtype=torch.long, device=dev); jj = torch.ones(1, dtype=torch.long, device=dev)
    ws = db.ba_workspace(1, P, F, ht * wd, dev)
    db.ba_plan(ii, jj, F, ht * wd, -1, 1, P + 1, ws)
    sysd = torch.from_numpy(sysm).to(dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    dx, _ = db.ba_finish(poses, disps, sysd, ii, jj, 1, P + 1, lm, ep, True, ws, status=status)
    torch.cuda.synchronize()
    zeroed = bool((sysd == 0).all())
    global LAST_US
    src = torch.from_numpy(sysm).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):                                   # the solve alone (the system is copied back in front of every call)
        sysd.copy_(src); poses[:, :6] = 0; poses[:, 6] = 1
        e0.record()
        db.ba_finish(poses, disps, sysd, ii, jj, 1, P + 1, lm, ep, True, ws)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    LAST_US = sorted(ts)[len(ts) // 2]
    return dx.cpu().numpy().reshape(-1).astype(np.float64), want, status.cpu().numpy(), zeroed


if __name__ == "__main__":
    if len(sys.argv) > 1:
        db.debug_config("ba_solver", sys.argv[1])
    worst = 0.0
    sizes = [int(x) for x in os.environ["SIZES"].split(",")] if os.environ.get("SIZES") else list(range(1, 33)) + [40, 63, 85, 128, 200]
    for P in sizes:
        got, want, status, zeroed = solve(P, seed=P)
        err = np.abs(got - want).max() / np.abs(want).max()
        worst = max(worst, err)
        print("P %3d  n %4d  rel err %.2e  status %s  sys zeroed %s  %8.1f us per solve (events, incl. launch gaps)" % (P, 6 * P, err, status.tolist(), zeroed, LAST_US))
    print("worst", worst)
