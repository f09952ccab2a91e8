"""Kernels of OTHER libraries beside this library's matrix-core kernels (include/pvo_hip.h "Streams").

Round 4 found that packed-FP32 VALU instructions (v_pk_fma_f32 ...) of one of this library's kernels returned wrong values in single
registers while a kernel of another stream that issues MFMA instructions was resident on the compute unit, and removed them from the
library (pvo_amd/build.py NO_PACKED_FP32; tests/test_c_abi.py disassembles the .so).  The converse exposure was only documented: a
CALLER's kernels - PyTorch's element-wise kernels are full of packed FP32 - running on another stream beside pvo_gru_conv_gates, as
they do in the training path and around MotionFilter.  This test runs exactly that, 2 000 times, and compares every result with the
serial run bit for bit, both ways (the caller's kernels and the library's)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_torch_elementwise_kernels_beside_the_gate_convolution_are_unharmed(cuda):
    from pvo_amd import droid_backends as db
    dev = torch.device(cuda)
    g = torch.Generator().manual_seed(11)
    E, H, W = 36, 48, 64
    net = torch.tanh(torch.randn(E, H, W, 128, generator=g)).half().to(dev).permute(0, 3, 1, 2)
    cf = torch.relu(torch.randn(E, H, W, 192, generator=g)).half().to(dev).permute(0, 3, 1, 2)
    w = (torch.randn(9, 256, 320, generator=g) * 0.02).half().to(dev)
    gg = torch.randn(E, 384, generator=g).to(dev)
    P = torch.randn(E, H, W, 256, generator=g).half().to(dev).permute(0, 3, 1, 2)
    # the caller's side: fp32 element-wise chains over 4 M values (vectorised float4 kernels: packed multiply-adds on gfx950)
    a, b, c = (torch.randn(4 << 20, generator=g).to(dev) for _ in range(3))

    def victim():
        x = torch.addcmul(c, a, b)              # c + a * b
        x = x * 1.0009765625 + a
        x = torch.lerp(x, b, 0.37)
        return (x * x + c) * b - a

    torch.cuda.synchronize()
    want_v = victim()
    want_z, want_rn = db.gru_conv_gates(net, cf, w, gg, P)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    bad_v = torch.zeros((), dtype=torch.int64, device=dev)
    bad_c = torch.zeros((), dtype=torch.int64, device=dev)
    reps = 2000
    for r in range(reps):
        z, rn = db.gru_conv_gates(net, cf, w, gg, P)               # ~170 us of MFMA on the current stream ...
        bad_c += (z != want_z).any() | (rn != want_rn).any()
        with torch.cuda.stream(side):
            for _ in range(3):                                     # ... and ~12 element-wise launches beside it; EVERY result is
                bad_v += (victim() != want_v).any()                # compared on the device, the counters are read once at the end
    torch.cuda.synchronize()
    assert int(bad_v) == 0, "%d of %d element-wise results differ from the serial run beside the gate convolution" % (int(bad_v), 3 * reps)
    assert int(bad_c) == 0, "%d of %d gate convolutions differ from the serial run beside element-wise kernels" % (int(bad_c), reps)
