"""numpy/ctypes front-end of oracle/libpvo_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  All arrays are host numpy
arrays; 16-bit types are carried as numpy float16 or as uint16 bit patterns (bf16).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpvo_oracle.so")

O_F32, O_F16, O_BF16, O_F64 = 0, 1, 2, 3
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            import sys
            sys.path.insert(0, os.path.dirname(_HERE))
            from pvo_amd.build import build_oracle
            build_oracle()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def dtype_code(a, bf16=False):
    if bf16:
        assert a.dtype == np.uint16
        return O_BF16
    return {np.dtype(np.float32): O_F32, np.dtype(np.float16): O_F16, np.dtype(np.float64): O_F64}[a.dtype]


def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7fffffff) > 0x7f800000
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def corr_index_forward(volume, coords, radius, contract=True, bf16=False):
    volume = np.ascontiguousarray(volume); coords = np.ascontiguousarray(coords, dtype=np.float32)
    N, h1, w1, h2, w2 = volume.shape
    rd = 2 * radius + 1
    out = np.zeros((N, rd, rd, h1, w1), dtype=volume.dtype)
    rc = lib().oracle_corr_index_forward(_p(volume), _p(coords), _p(out), N, h1, w1, h2, w2, radius,
                                         dtype_code(volume, bf16), int(contract))
    assert rc == 0
    return out


def corr_index_backward(volume_shape, coords, corr_grad, radius, contract=True, bf16=False):
    coords = np.ascontiguousarray(coords, dtype=np.float32); corr_grad = np.ascontiguousarray(corr_grad)
    N, h1, w1, h2, w2 = volume_shape
    out = np.zeros(volume_shape, dtype=corr_grad.dtype)
    rc = lib().oracle_corr_index_backward(_p(coords), _p(corr_grad), _p(out), N, h1, w1, h2, w2, radius,
                                          dtype_code(corr_grad, bf16), int(contract))
    assert rc == 0
    return out


def corr_pyramid_lookup(pyramid, coords_nhw2, radius, contract=True, bf16=False):
    pyramid = [np.ascontiguousarray(v) for v in pyramid]
    coords = np.ascontiguousarray(coords_nhw2, dtype=np.float32)
    N, h1, w1, h2, w2 = pyramid[0].shape
    L = len(pyramid)
    rd = 2 * radius + 1
    out = np.zeros((N, L * rd * rd, h1, w1), dtype=pyramid[0].dtype)
    ptrs = (ctypes.c_void_p * L)(*[v.ctypes.data for v in pyramid])
    rc = lib().oracle_corr_pyramid_lookup(ptrs, _p(coords), _p(out), N, h1, w1, h2, w2, L, radius,
                                          dtype_code(pyramid[0], bf16), int(contract))
    assert rc == 0
    return out


def corr_build(fmap1, fmap2, num_levels=4, bf16=False):
    fmap1 = np.ascontiguousarray(fmap1); fmap2 = np.ascontiguousarray(fmap2)
    N, C, H, W = fmap1.shape
    levels = [np.zeros((N, H, W, H >> l, W >> l), dtype=fmap1.dtype) for l in range(num_levels)]
    ptrs = (ctypes.c_void_p * num_levels)(*[v.ctypes.data for v in levels])
    rc = lib().oracle_corr_build(_p(fmap1), _p(fmap2), ptrs, N, C, H, W, num_levels, dtype_code(fmap1, bf16))
    assert rc == 0
    return levels


# --------------------------------------------------------------------------- geometry / BA
def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64c(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    poses, disps, intrinsics, ii, jj = _f32c(poses), _f32c(disps), _f32c(intrinsics), _i64c(ii), _i64c(jj)
    out = np.zeros(ii.shape[0], np.float32)
    lib().oracle_frame_distance(_p(poses), _p(disps), _p(intrinsics), _p(ii), _p(jj), _p(out),
                                int(ii.shape[0]), disps.shape[1], disps.shape[2], ctypes.c_float(beta))
    return out


def projmap(poses, disps, intrinsics, ii, jj):
    poses, disps, intrinsics, ii, jj = _f32c(poses), _f32c(disps), _f32c(intrinsics), _i64c(ii), _i64c(jj)
    E, ht, wd = ii.shape[0], disps.shape[1], disps.shape[2]
    coords = np.zeros((E, ht, wd, 3), np.float32); valid = np.zeros((E, ht, wd, 1), np.float32)
    lib().oracle_projmap(_p(poses), _p(disps), _p(intrinsics), _p(ii), _p(jj), _p(coords), _p(valid), E, ht, wd)
    return coords, valid


def iproj(poses, disps, intrinsics):
    poses, disps, intrinsics = _f32c(poses), _f32c(disps), _f32c(intrinsics)
    N, ht, wd = disps.shape
    pts = np.zeros((N, ht, wd, 3), np.float32)
    lib().oracle_iproj(_p(poses), _p(disps), _p(intrinsics), _p(pts), N, ht, wd)
    return pts


def depth_filter(poses, disps, intrinsics, ix, thresh):
    poses, disps, intrinsics, ix, thresh = _f32c(poses), _f32c(disps), _f32c(intrinsics), _i64c(ix), _f32c(thresh)
    N, (nf, ht, wd) = ix.shape[0], disps.shape
    out = np.zeros((N, ht, wd), np.float32)
    lib().oracle_depth_filter(_p(poses), _p(disps), _p(intrinsics), _p(ix), _p(thresh), _p(out), N, nf, ht, wd)
    return out


def reproject(poses, disps, intrinsics, ii, jj):
    poses, disps, intrinsics, ii, jj = _f32c(poses), _f32c(disps), _f32c(intrinsics), _i64c(ii), _i64c(jj)
    E, ht, wd = ii.shape[0], disps.shape[1], disps.shape[2]
    coords = np.zeros((E, ht, wd, 2), np.float32); valid = np.zeros((E, ht, wd, 1), np.float32)
    lib().oracle_reproject(_p(poses), _p(disps), _p(intrinsics), _p(ii), _p(jj), _p(coords), _p(valid), E, ht, wd)
    return coords, valid


def ba(poses, disps, intrinsics, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
       motion_only=False, xi45_zero=False, evt_skip_first=True, want_sys=False):
    """Returns dict(poses, disps, dx, dz, K, failed[, sys]); inputs are not modified."""
    poses, disps = _f32c(poses).copy(), _f32c(disps).copy()
    intrinsics, targets, weights = _f32c(intrinsics), _f32c(targets), _f32c(weights)
    ii, jj = _i64c(ii), _i64c(jj)
    nf, ht, wd = disps.shape
    E, P = ii.shape[0], t1 - t0
    if eta is None:
        eta = np.zeros((1, ht, wd), np.float32)
    eta = _f32c(eta).reshape(-1, ht, wd)
    dx = np.zeros((max(P, 0), 6), np.float32)
    dz = np.zeros((P + E + 1, ht * wd), np.float32)
    sys_ = np.zeros((6 * P) * (6 * P) + 6 * P, np.float64)
    status = np.zeros(4, np.int32)
    f = lib().oracle_ba
    f.restype = ctypes.c_int
    K = f(_p(poses), _p(disps), _p(intrinsics), _p(targets), _p(weights), _p(eta), _p(ii), _p(jj),
          E, nf, ht, wd, eta.shape[0], t0, t1, iterations, ctypes.c_float(lm), ctypes.c_float(ep),
          int(motion_only), _p(dx), _p(dz), _p(sys_) if want_sys else None, _p(status),
          int(xi45_zero), int(evt_skip_first), None)
    if K < 0:
        raise ValueError("oracle_ba failed with %d" % K)
    out = dict(poses=poses, disps=disps, dx=dx, dz=dz[:K], K=K, failed=bool(status[0]))
    if want_sys:
        out["sys"] = sys_
    return out


def pose_retr(poses, dx, t0, t1, xi45_zero=False):
    """pose_retr_kernel (droid_kernels.cu:877-910): poses[k] <- retrSE3(dx[k - t0], poses[k]) for k in [t0, t1); xi45_zero
    reproduces "the out-of-bounds read xi[45] of expSE3 (:154) returned 0", the default is upstream's xi[5]"""
    out = _f32c(poses).copy()
    dx = _f32c(dx)
    for k in range(t0, t1):
        p1 = np.zeros(7, np.float32)
        lib().oracle_retrSE3(_p(np.ascontiguousarray(dx[k - t0])), _p(np.ascontiguousarray(out[k])), _p(p1), int(xi45_zero))
        out[k] = p1
    return out


def ba_assemble(poses, disps, intrinsics, targets, weights, ii, jj):
    """projective_transform_kernel alone (droid_kernels.cu:177-403): dict(Hs, vs [fp64 sums], Eii, Eij, Cii, bz [fp32])"""
    poses, disps, intrinsics = _f32c(poses), _f32c(disps), _f32c(intrinsics)
    targets, weights, ii, jj = _f32c(targets), _f32c(weights), _i64c(ii), _i64c(jj)
    (_, ht, wd), E = disps.shape, ii.shape[0]
    HW = ht * wd
    Hs, vs = np.zeros((4, E, 6, 6), np.float64), np.zeros((2, E, 6), np.float64)
    Eii, Eij = np.zeros((E, 6, HW), np.float32), np.zeros((E, 6, HW), np.float32)
    Cii, bz = np.zeros((E, HW), np.float32), np.zeros((E, HW), np.float32)
    lib().oracle_ba_assemble(_p(poses), _p(disps), _p(intrinsics), _p(targets), _p(weights), _p(ii), _p(jj), E, ht, wd,
                             _p(Hs), _p(vs), _p(Eii), _p(Eij), _p(Cii), _p(bz))
    return dict(Hs=Hs, vs=vs, Eii=Eii, Eij=Eij, Cii=Cii, bz=bz)


def ba_apply(poses, disps, intrinsics, targets, weights, eta, ii, jj, t0, t1, dx, motion_only=False):
    """one BA iteration on these edges with the pose update `dx` imposed (sharded-BA test harness)"""
    poses, disps = _f32c(poses).copy(), _f32c(disps).copy()
    intrinsics, targets, weights = _f32c(intrinsics), _f32c(targets), _f32c(weights)
    ii, jj = _i64c(ii), _i64c(jj)
    nf, ht, wd = disps.shape
    E, P = ii.shape[0], t1 - t0
    if eta is None:
        eta = np.zeros((1, ht, wd), np.float32)
    eta = _f32c(eta).reshape(-1, ht, wd)
    dxo = np.zeros((max(P, 0), 6), np.float32); dz = np.zeros((P + E + 1, ht * wd), np.float32)
    status = np.zeros(4, np.int32)
    dxi = _f32c(dx)
    f = lib().oracle_ba
    f.restype = ctypes.c_int
    K = f(_p(poses), _p(disps), _p(intrinsics), _p(targets), _p(weights), _p(eta), _p(ii), _p(jj),
          E, nf, ht, wd, eta.shape[0], t0, t1, 1, ctypes.c_float(0.0), ctypes.c_float(1.0),
          int(motion_only), _p(dxo), _p(dz), None, _p(status), 0, 1, _p(dxi))
    assert K >= 0
    return dict(poses=poses, disps=disps, dz=dz[:K])


def altcorr_forward(fmap1, fmap2, coords, radius):
    fmap1, fmap2, coords = _f32c(fmap1), _f32c(fmap2), _f32c(coords)
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    out = np.zeros((B, S, rd * rd, H1, W1), np.float32)
    assert lib().oracle_altcorr_forward(_p(fmap1), _p(fmap2), _p(coords), _p(out), B, S, H1, W1, H2, W2, C, radius) == 0
    return out


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    fmap1, fmap2, coords, corr_grad = _f32c(fmap1), _f32c(fmap2), _f32c(coords), _f32c(corr_grad)
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    g1, g2 = np.zeros_like(fmap1), np.zeros_like(fmap2)
    assert lib().oracle_altcorr_backward(_p(fmap1), _p(fmap2), _p(coords), _p(corr_grad), _p(g1), _p(g2),
                                         B, S, H1, W1, H2, W2, C, radius) == 0
    return g1, g2
