"""BasicEncoder.forward_inference (pvo_bias_norm_act: bias + instance norm + ReLU + residual add as one kernel per layer) against
the module's own forward under fp16 autocast - the path the reference runs per frame (motion_filter.py:52-60, extractor.py:6-56,
183-201) - and the kernel alone against the ATen operations it replaces, step by step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 32, 120, 404), (1, 64, 60, 202), (3, 128, 30, 101), (2, 16, 7, 9)])
@pytest.mark.parametrize("norm,residual", [(True, False), (True, True), (False, True), (False, False)])
def test_bias_norm_act_equals_the_separate_operations(cuda, shape, norm, residual):
    from pvo_amd import droid_backends as db
    g = torch.Generator().manual_seed(shape[1] + shape[2])
    x = (torch.randn(shape, generator=g) * 2.0 + 0.3).half().to(cuda)
    b = torch.randn(shape[1], generator=g).half().to(cuda)
    r = torch.randn(shape, generator=g).half().to(cuda) if residual else None
    got = db.bias_norm_act(x, b, r, norm=norm, relu_inner=True, relu_outer=residual)
    t = x + b.view(1, -1, 1, 1)                                              # fp16 add, as `conv(x) + bias` rounds
    if norm:
        t = torch.nn.functional.instance_norm(t)                             # fp32 statistics, fp16 output
    t = torch.relu(t)
    if residual:
        t = torch.relu(r + t)
    # identical except where the two-pass statistics and batch_norm's Welford pass differ in the last fp32 bits: one fp16 ulp
    d = (got.float() - t.float()).abs()
    assert float(d.max()) <= 2e-3 * max(float(t.float().abs().max()), 1.0)
    assert float((d > 0).float().mean()) < (0.02 if norm else 1e-9)
    if not norm:
        assert torch.equal(got, t)


@pytest.mark.parametrize("norm_fn,out_dim", [("instance", 128), ("none", 256)])
def test_fused_encoder_equals_the_module_under_autocast(cuda, norm_fn, out_dim):
    from pvo_amd.modules.extractor import BasicEncoder
    torch.manual_seed(3)
    enc = BasicEncoder(output_dim=out_dim, norm_fn=norm_fn).to(cuda).eval().half()
    g = torch.Generator().manual_seed(5)
    for (h, w) in ((240, 808), (64, 96)):
        x = torch.randn(1, 1, 3, h, w, generator=g).to(cuda)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            want = enc(x)
            got = enc.forward_inference(x)
        assert got.shape == want.shape and got.dtype == want.dtype
        scale = float(want.float().abs().max())
        err = float((got.float() - want.float()).abs().max())
        print(norm_fn, (h, w), "max |diff| %.3g of scale %.3g" % (err, scale))
        assert err <= 2e-2 * scale                                            # 13 layers of fp16 roundings: ulp-level differences spread
