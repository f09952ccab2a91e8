"""Which buffer differs first when the BA runs beside other kernels of the update?   (on the GPU box)

    python tools/sched_bisect.py --build            # here: tools/_probe/libpvo_hip_sched.so  (-DPVO_SCHED_DEBUG)
    python tools/sched_bisect.py [runs]             # on the GPU

Round 2 found two stream arrangements of pvo_graph_update whose poses were not bitwise reproducible:
the BA beside the upsampling-mask convolution (1 of ~290 two-update runs) and the first edge-block assembly launched before
the wait on the eta head (13 of 13).  The debug build selects the arrangement (pvo_debug_sched: 0 shipped, 1 BA beside the
mask convolution, 2 = 1 + assembly before the wait, 3 mask convolution after the BA) and copies the pose system and the whole
BA workspace after every assembly+Schur and after every solve+back-substitution into a tap; runs are repeated from one fixed
state and compared bit for bit with the first run of the same arrangement: final state, then the taps in order, and inside
the first differing tap the named parts of the workspace (Eii, Eij, Cii, bz | Ei, Q, w | dx ...) and the BA's inputs.
"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")
# --variant pk (or PVO_SCHED_VARIANT=pk): the library built WITH packed-FP32 VALU instructions, i.e. as it was before round 4 -
# the build in which arrangements 1 / 2 / 7 / 9 differ from run to run (profiles/r04_coresidency.md).  Default: the shipped flags.
# (Round 4 also had variants with agent-scope loads / stores of the BA's buffers, explicit fences and a ds_bpermute wave sum; their
# switches are gone from the sources, their results are in that file.)
VARIANT = os.environ.get("PVO_SCHED_VARIANT", "")
for k, a_ in enumerate(sys.argv):
    if a_ == "--variant":
        VARIANT = sys.argv[k + 1]
        del sys.argv[k:k + 2]
        break
assert VARIANT in ("", "pk")
PROBE_LIB = os.path.join(PROBE_DIR, "libpvo_hip_sched%s.so" % (("_" + VARIANT) if VARIANT else ""))
if "--build" in sys.argv:
    from pvo_amd import build
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    flags = [f for f in build.HIPCC_FLAGS]
    if VARIANT == "pk":
        k = flags.index(build.NO_PACKED_FP32[0])
        del flags[k:k + len(build.NO_PACKED_FP32)]
    objs = []
    for s in build.HIP_SOURCES:
        if s in ("update_exec.hip", "ba.hip", "graph_glue.hip") or VARIANT == "pk":
            obj = os.path.join(PROBE_DIR, "sched%s_" % VARIANT + s.replace(".hip", ".o"))
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + (["-DPVO_SCHED_DEBUG"] if s in ("update_exec.hip", "ba.hip", "graph_glue.hip") else []) +
                                  ["-c", os.path.join(build.CSRC, s), "-o", obj], stderr=subprocess.DEVNULL)
        else:
            obj = os.path.join(build.CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_LIB] + objs)
    print(PROBE_LIB); sys.exit(0)

import torch
from pvo_amd import _lib
_lib.LIB_PATH = PROBE_LIB
import bench                                           # noqa: E402
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if not a.startswith("--")]
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N_UPD = int(os.environ.get("PVO_CHECK_UPDATES", "2"))
ITRS = int(os.environ.get("PVO_CHECK_ITRS", "2"))
# mode[:flags]  flags: s = events without the system-scope fence (hipEventDisableSystemFence), d = device-scope release
# (hipEventReleaseToDevice), n = mode 1/2 without the event record on the side stream behind the mask convolution
# h = "fence hammer": while the updates run, a third stream executes a train of default-flag event records (each a marker with a
# system-scope cache writeback + invalidate) and nothing else
MODES = os.environ.get("PVO_SCHED_MODES", "0,0:s,1:s,0:h,0:sh,1:sh").split(",")
EVF = {"": 0x2, "s": 0x2 | 0x20000000, "d": 0x2 | 0x40000000, "n": 0x2, "h": 0x2, "sh": 0x2 | 0x20000000}
HAMMER = False
dev = torch.device("cuda:0")
hammer_stream = torch.cuda.Stream(dev)
lib = _lib.load()
lib.pvo_debug_sched.restype = ctypes.c_int
lib.pvo_debug_sched.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
lib.pvo_debug_sched_taps.restype = ctypes.c_int
lib.pvo_debug_event_flags.restype = ctypes.c_int
lib.pvo_debug_event_flags.argtypes = [ctypes.c_uint]
lib.pvo_debug_ba_layout.restype = ctypes.c_int
lib.pvo_debug_ba_layout.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]

video, graph = bench.make_window(dev, seed=0)
names = ("net", "target_cam", "weight", "raw_mask", "delta_dy")
state = {n: getattr(graph, n).clone() for n in names}
p0, d0, dm0 = video.poses.clone(), video.disps.clone(), graph.damping.clone()
graph.update(None, None, use_inactive=True)            # builds the per-edge-set cache (BA plan, workspaces)
torch.cuda.synchronize()
st = graph._cache["fused"]
ba = st["ba"]
E_ba, P = int(st["ii_ba"].shape[0]), None
a = st["args"]
P = a.t1 - a.t0
F, HW = video.disps.shape[0], graph.ht * graph.wd
lay = (ctypes.c_size_t * 16)()
assert lib.pvo_debug_ba_layout(E_ba, P, F, HW, lay) == 0
PARTS = ["kidx", "kx", "eptr", "eidx", "meta", "env", "Eii", "Eij", "Cii", "bz", "Ei", "Q", "w", "dx", "sys(ws)"]
offs = list(lay)[:15] + [lay[15]]
n6 = 6 * P
sys_bytes = 8 * (n6 * n6 + n6)
sys_pad = (sys_bytes + 255) & ~255
ws_bytes = ba["ws"].numel()
ws_skew = (-ba["ws"].data_ptr()) % 256                  # the library aligns the workspace base up to 256 bytes
slot = (sys_pad + ws_bytes + 255) & ~255
n_slots = 2 * 2 * N_UPD                                 # (itrs = 2) x (after local, after finish) x updates
tap = torch.zeros(slot * n_slots, dtype=torch.uint8, device=dev)
STAGE = ["assemble+schur", "solve+backsub"]


def run(mode, with_tap=True):
    for n in names:
        getattr(graph, n).copy_(state[n])
    video.poses.copy_(p0); video.disps.copy_(d0); graph.damping.copy_(dm0)
    tap.zero_()
    torch.cuda.synchronize()
    lib.pvo_debug_sched(mode, tap.data_ptr() if with_tap else None, tap.numel(), slot)
    if HAMMER:
        evs = [torch.cuda.Event() for _ in range(int(os.environ.get("PVO_HAMMER_N", "400")))]
        hammer_stream.wait_stream(torch.cuda.current_stream(dev))
        for e in evs:
            e.record(hammer_stream)
    for _ in range(N_UPD):
        graph.update(None, None, itrs=ITRS, use_inactive=True)
    torch.cuda.synchronize()
    out = dict(net=graph.net.clone(), target=graph.target_cam.clone(), weight=graph.weight.clone(), raw_mask=graph.raw_mask.clone(),
               damping=graph.damping.clone(), poses=video.poses.clone(), disps=video.disps.clone(),
               target_ba=st["target_ba"].clone(), weight_ba=st["weight_ba"].clone())
    return out, (tap.clone() if with_tap else None)


def bits(t):
    return t.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()])


def first_diff(ref_tap, got_tap):
    for k in range(n_slots):
        r, g = ref_tap[k * slot:(k + 1) * slot], got_tap[k * slot:(k + 1) * slot]
        if torch.equal(r, g):
            continue
        upd, it, stage = k // 4, (k // 2) % 2, k % 2
        what = []
        if not torch.equal(r[:sys_bytes], g[:sys_bytes]):
            d = (r[:sys_bytes].view(torch.int64) != g[:sys_bytes].view(torch.int64)).nonzero().flatten()
            what.append("sys (%d of %d entries, first %d = row %d col %d)" % (d.numel(), sys_bytes // 8, d[0], d[0] // n6, d[0] % n6))
        wr, wg = r[sys_pad + ws_skew:sys_pad + ws_bytes], g[sys_pad + ws_skew:sys_pad + ws_bytes]
        for i, name in enumerate(PARTS):
            lo, hi = offs[i], offs[i + 1]
            if hi <= wr.numel() and not torch.equal(wr[lo:hi], wg[lo:hi]):
                d = (wr[lo:hi].view(torch.int32) != wg[lo:hi].view(torch.int32)).nonzero().flatten()
                fr, fg = wr[lo:hi].view(torch.float32)[d[:4]], wg[lo:hi].view(torch.float32)[d[:4]]
                what.append("%s (%d words; first at %d: %s vs %s)" % (name, d.numel(), d[0], fr.tolist(), fg.tolist()))
        return "update %d, iteration %d, after %s: %s" % (upd, it, STAGE[stage], "; ".join(what) or "padding only")
    return None


TIMES = os.environ.get("PVO_SCHED_TIMES") == "1"
if TIMES:
    # per run: a log of constant-rate clock stamps (100 MHz) - graph_post workgroups when their stores are acknowledged (tag 1),
    # assembly workgroups when they start (tag 2).  In stream order every assembly of an update starts after the LAST graph_post
    # workgroup of that update has finished; the first assembly of an update is the first kernel behind the wait on `mid`.
    lib.pvo_debug_log_graph.argtypes = [ctypes.c_void_p]; lib.pvo_debug_log_ba.argtypes = [ctypes.c_void_p]
    tlog = torch.zeros(16002, dtype=torch.int64, device=dev)
    assert lib.pvo_debug_log_graph(tlog.data_ptr()) == 0 and lib.pvo_debug_log_ba(tlog.data_ptr()) == 0
    for spec in MODES:
        mode, _, fl = spec.partition(":")
        assert lib.pvo_debug_event_flags(EVF[fl]) == 0
        run(int(mode), False)
        ref, _ = run(int(mode), False)
        early_runs = bad = 0; worst = 0.0; gaps = []
        for r in range(RUNS):
            tlog.zero_(); torch.cuda.synchronize()
            out, _ = run(int(mode), False)
            differs = any(not torch.equal(bits(out[k]), bits(ref[k])) for k in ref)
            n = int(tlog[0]); ent = tlog[1:1 + n].cpu().numpy()
            tag, t = (ent >> 62) & 3, ent & ((1 << 62) - 1)
            order = t.argsort(kind="stable"); tag, t = tag[order], t[order]
            early = 0.0
            # per update (by counts, in clock order): graph_post's workgroups, then the two assemblies'
            posts, asms = t[tag == 1], t[tag == 2]
            npost = posts.size // N_UPD; nasm = asms.size // N_UPD
            for u in range(N_UPD):
                p_u = np.sort(posts)[u * npost:(u + 1) * npost]; a_u = np.sort(asms)[u * nasm:(u + 1) * nasm]
                gap = (a_u.min() - p_u.max()) * 0.01          # us: first assembly start minus last graph_post end
                gaps.append(gap); early = min(early, gap)
            early_runs += early < 0; bad += differs; worst = min(worst, early)
            if (early < 0 or differs) and early_runs + bad <= 8:
                print("mode %s run %d: result %s; first assembly start - last graph_post end = %.2f us (negative = the assembly STARTED BEFORE graph_post FINISHED)" % (spec, r, "DIFFERS" if differs else "equal", early), flush=True)
        g = np.array(gaps)
        print("mode %s: %d of %d runs differ; %d runs with an assembly workgroup that started before the last graph_post workgroup finished (worst %.2f us); gap median %.2f us, min %.2f us" % (
            spec, bad, RUNS, early_runs, worst, float(np.median(g)), float(g.min())), flush=True)
    sys.exit(0)

if os.environ.get("PVO_SCHED_PARTIALS") == "1":
    # The assembly's chunk sums: every wave's 90 sums before they cross LDS (debug copy) and the workgroup's sums after (`part`).
    # One update with ONE BA iteration per run (PVO_CHECK_UPDATES=1 PVO_CHECK_ITRS=1), so that the first tap and the debug copy
    # belong to the assembly that runs beside the side stream's kernel; the clean reference is the shipped arrangement (mode 0).
    assert N_UPD == 1 and ITRS == 1
    lib.pvo_debug_partials.argtypes = [ctypes.c_void_p]
    chunks = (HW + 511) // 512
    pbuf = torch.zeros(E_ba * chunks * 4 * 90, dtype=torch.float32, device=dev)
    assert lib.pvo_debug_partials(pbuf.data_ptr()) == 0
    part_lo = offs[PARTS.index("dx")] // 4 + ((4 * (6 * P + 8) + 255) // 256) * 64           # `part` follows dx (256-byte aligned)
    def one(mode):
        pbuf.zero_(); torch.cuda.synchronize()
        out, t = run(mode)
        ws = t[0:slot][sys_pad + ws_skew:sys_pad + ws_bytes].view(torch.float32)
        return out, pbuf.clone().view(E_ba, chunks, 4, 90), ws[part_lo:part_lo + E_ba * chunks * 90].clone().view(E_ba, chunks, 90)
    assert lib.pvo_debug_event_flags(EVF[""]) == 0
    one(0); ref_out, ref_pw, ref_part = one(0)
    _, pw2, part2 = one(0)
    summed = lambda pw: (pw[:, :, 0] + pw[:, :, 1]) + (pw[:, :, 2] + pw[:, :, 3])
    print("mode 0 twice: partials equal %s, part equal %s; part == (w0 + w1) + (w2 + w3) of the partials: %s" % (
        torch.equal(pw2, ref_pw), torch.equal(part2, ref_part), torch.equal(bits(summed(ref_pw)), bits(ref_part))))
    mode = int(MODES[-1].split(":")[0])
    nbad = 0
    for r in range(RUNS):
        out, pw, part = one(mode)
        dpart = (bits(part) != bits(ref_part)).nonzero()
        dpw = (bits(pw) != bits(ref_pw)).nonzero()
        lds = (bits(summed(pw)) != bits(part)).nonzero()
        if dpart.numel() == 0 and dpw.numel() == 0:
            continue
        nbad += 1
        if nbad > 10:
            continue
        print("mode %d run %d: %d of %d chunk sums differ from the clean run; %d of %d WAVE partials differ from the clean run; %d chunk sums are not the sum of this run's own four partials" % (
            mode, r, dpart.shape[0], ref_part.numel(), dpw.shape[0], ref_pw.numel(), lds.shape[0]), flush=True)
        for e_, c_, w_, l_ in dpw[:10].tolist():
            print("      edge %d chunk %d wave %d sum %d: partial %.9g (clean %.9g)" % (e_, c_, w_, l_, float(pw[e_, c_, w_, l_]), float(ref_pw[e_, c_, w_, l_])))
        waves = sorted(set((e_, c_, w_) for e_, c_, w_, l_ in dpw.tolist()))
        print("      waves hit: %s" % ", ".join("(edge %d chunk %d wave %d: sums %s)" % (e_, c_, w_, [l for a1, a2, a3, l in dpw.tolist() if (a1, a2, a3) == (e_, c_, w_)]) for e_, c_, w_ in waves[:8]))
    print("mode %d: %d of %d runs differ" % (mode, nbad, RUNS))
    sys.exit(0)

if os.environ.get("PVO_SCHED_DETAIL") == "1":
    # where exactly do the first differing taps differ?  decode every differing word of Eii / Eij / Cii / bz / part into (edge, row,
    # pixel) and test it against the SAME word one tap earlier in this run and in the reference run (is it an old value?)
    spec = MODES[-1]
    mode, _, fl = spec.partition(":")
    assert lib.pvo_debug_event_flags(EVF[fl]) == 0
    run(int(mode)); ref, ref_tap = run(int(mode))
    shown = 0
    for r in range(RUNS):
        out, t = run(int(mode))
        for k in range(n_slots):
            rs, gs = ref_tap[k * slot:(k + 1) * slot], t[k * slot:(k + 1) * slot]
            if torch.equal(rs, gs) or k % 2 == 1:
                continue
            wr, wg = rs[sys_pad + ws_skew:sys_pad + ws_bytes], gs[sys_pad + ws_skew:sys_pad + ws_bytes]
            msgs = []
            for i, name in enumerate(PARTS):
                lo, hi = offs[i], offs[i + 1]
                if name not in ("Eii", "Eij", "Cii", "bz", "Ei", "Q", "w", "dx") or hi > wr.numel() or torch.equal(wr[lo:hi], wg[lo:hi]):
                    continue
                a_, b_ = wr[lo:hi].view(torch.int32), wg[lo:hi].view(torch.int32)
                d = (a_ != b_).nonzero().flatten().cpu().numpy()
                old = None
                if k >= 2:        # the same words after the previous assemble+schur of THIS run
                    prev = t[(k - 2) * slot:(k - 1) * slot][sys_pad + ws_skew:sys_pad + ws_bytes][lo:hi].view(torch.int32)
                    old = int((prev[torch.from_numpy(d).to(dev)] == b_[torch.from_numpy(d).to(dev)]).sum())
                sect = np.unique(d // 16)
                if name in ("Eii", "Eij"):
                    where = sorted(set((int(x) // (6 * HW), (int(x) // HW) % 6, (int(x) % HW) // 16 * 16) for x in d))
                    by_px = {}
                    for e_, n_, p_ in where:
                        by_px.setdefault((e_, p_), []).append(n_)
                    desc = "; ".join("edge %d px %d..%d rows %s" % (e_, p_, p_ + 15, rows) for (e_, p_), rows in sorted(by_px.items())[:6])
                elif name in ("Cii", "bz"):
                    desc = "; ".join("edge %d px %d" % (int(x) // HW, int(x) % HW) for x in d[:6])
                elif name == "dx":
                    dd = d[d >= 6 * P + 8] - (6 * P + 8)
                    desc = "part: " + "; ".join("edge %d chunk %d sum %d" % (int(x) // 540, (int(x) // 90) % 6, int(x) % 90) for x in dd[:24]) + (" | dx itself: %d words" % int((d < 6 * P + 8).sum()))
                else:
                    desc = "words %s" % d[:8].tolist()
                msgs.append("%s: %d words in %d 64-byte sectors (%s of them equal this run's previous assemble+schur tap) [%s]" % (name, d.size, sect.size, old, desc))
            print("run %d tap %d (update %d iteration %d):\n    %s" % (r, k, k // 4, (k // 2) % 2, "\n    ".join(msgs) or "other parts only"), flush=True)
            shown += 1
            break
        if shown >= 12:
            break
    sys.exit(0)

for spec in MODES:
    mode, _, fl = spec.partition(":")
    mode = int(mode) | (256 if fl == "n" else 0)
    HAMMER = "h" in fl
    assert lib.pvo_debug_event_flags(EVF[fl]) == 0
    run(mode)
    ref, ref_tap = run(mode)
    bad = 0
    for r in range(RUNS):
        out, t = run(mode)
        diff = [k for k in ref if not torch.equal(bits(out[k]), bits(ref[k]))]
        fd = first_diff(ref_tap, t)
        if fd and fd.endswith(": dx (42 words" + fd.split(": dx (42 words")[-1]) and "; " not in fd.split("after ")[-1] and "assemble" in fd:
            fd = None          # (dx in front of the first solve of a run is the PREVIOUS run's: not a difference of this run)
        if diff or fd:
            bad += 1
            if bad <= 6:
                print("mode %s run %d: final state differs in [%s]; first tap: %s" % (
                    spec, r, ", ".join("%s (max %.3g)" % (k, (out[k].float() - ref[k].float()).abs().max().item()) for k in diff), fd), flush=True)
    # the same arrangement WITHOUT the tap copies (they change what runs beside what)
    ref2, _ = run(mode, False)
    bad2 = 0
    for r in range(RUNS):
        out, _ = run(mode, False)
        bad2 += any(not torch.equal(bits(out[k]), bits(ref2[k])) for k in ref2)
    # ... and against the shipped arrangement: the arrangement must not change a bit either
    print("mode %s: %d of %d runs differ with the tap, %d of %d without" % (spec, bad, RUNS, bad2, RUNS), flush=True)
    if spec == MODES[0]:
        base = ref2
    else:
        d = [k for k in base if not torch.equal(bits(ref2[k]), bits(base[k]))]
        print("mode %s vs mode %s (no tap): %s" % (spec, MODES[0], "bitwise equal" if not d else "DIFFERS in " + ", ".join(d)), flush=True)
