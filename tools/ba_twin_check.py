"""The partitioned pose solve against the one-chain solve: partition chosen, agreement, time per BA call.
    python tools/ba_twin_check.py            (runs itself once per solver: the choice is read once per process)
    NF=64 HT=8 WD=10 python tools/ba_twin_check.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

def child(path):
    from pvo_amd import droid_backends as db
    db.debug_config("ba_solver", os.environ.get("TOOL_BA_SOLVER"))          # (the library's test hook; this tool's own variable)
    from test_geom_ba_gpu import _scene
    nf, ht, wd = int(os.environ.get("NF", "64")), int(os.environ.get("HT", "8")), int(os.environ.get("WD", "10"))
    s = _scene(7, nf, ht, wd, 3, 1)
    d = lambda t: t.cuda()
    args = [d(s[k]) for k in ("intr", "target", "weight", "eta", "ii", "jj")]
    st = torch.zeros(4, dtype=torch.int32, device="cuda")
    def run(its=2):
        p, q = d(s["poses"].clone()), d(s["disps"].clone())
        dx, dz = db.ba(p, q, *args, s["t0"], s["t1"], its, 1e-4, 0.1, False, status=st)
        return p, q, dx
    dx1 = run(1)[2]
    p, q, dx = run()
    part = db.ba_last_partition(args[4].shape[0], s["t1"] - s["t0"], nf, ht * wd, "cuda")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ps, qs = [d(s["poses"].clone()) for _ in range(20)], [d(s["disps"].clone()) for _ in range(20)]
    e0.record()
    for k in range(20):
        db.ba(ps[k], qs[k], *args, s["t0"], s["t1"], 2, 1e-4, 0.1, False)
    e1.record(); torch.cuda.synchronize()
    torch.save(dict(p=p.cpu(), q=q.cpu(), dx=dx.cpu(), dx1=dx1.cpu(), part=part, ms=e0.elapsed_time(e1) / 20, st=st.cpu()), path)

if len(sys.argv) > 1:
    child(sys.argv[1]); sys.exit(0)
out = {}
for solver in ("pipe", "twin"):
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), f.name], env=dict(os.environ, TOOL_BA_SOLVER=solver), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        if r.returncode != 0:
            print(solver, "FAILED\n", r.stdout[-3000:]); sys.exit(1)
        out[solver] = torch.load(f.name)
a, b = out["pipe"], out["twin"]
print("poses %d: partition (m, s) = %s | status pipe %s twin %s" % (a["p"].shape[0], b["part"], a["st"].tolist(), b["st"].tolist()))
print("  2 Gauss-Newton steps: one chain %.3f ms, partitioned %.3f ms" % (a["ms"], b["ms"]))
print("  ONE step (the same system into both solvers): max |dx| %.3e, difference %.3e" % (a["dx1"].abs().max(), (a["dx1"] - b["dx1"]).abs().max()))
print("  two steps (the second system is built from the first step's fp32 poses): max |dx| %.3e; differences: dx %.3e  poses %.3e  disps %.3e" % (a["dx"].abs().max(), (a["dx"] - b["dx"]).abs().max(), (a["p"] - b["p"]).abs().max(), (a["q"] - b["q"]).abs().max()))
