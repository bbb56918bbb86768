"""CPU: the quad-convolution identities and the index conventions of csrc/conv_q.h / wgrad_q.h / conv_q.hip (restated in tests/quad_ref.py)
against torch's conv2d + avg_pool2d / interpolate in fp64 -- forward, data gradient (through the transformed flipped image) and weight
gradient (through the fold) of both forms. Reference ops replaced: src/models/big_resnet.py:28-42 (F.interpolate + conv2d1),
:177-192,221-242 (conv2d2 + average_pooling)."""
import pytest
import torch

import quad_ref as Q


@pytest.mark.parametrize("H,W", [(4, 4), (8, 4), (6, 10)])
def test_quad_forms_match_torch(H, W):
    g = torch.Generator().manual_seed(5)
    N, C, Co = 2, 3, 4
    w9 = torch.randn(Co, 3, 3, C, generator=g, dtype=torch.float64)
    # POOL: x fine [N, 2H, 2W, C]
    x = torch.randn(N, 2 * H, 2 * W, C, generator=g, dtype=torch.float64).requires_grad_(True)
    wt = w9.clone().requires_grad_(True)
    y = Q.pool_conv_torch(x, wt)
    yq = Q.convq_ref(x.detach(), Q.quad_pack_ref(w9, 0), 0)
    assert torch.allclose(y.detach(), yq, atol=1e-12)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    dx, dw = torch.autograd.grad(y, (x, wt), dy)
    dxq = Q.convq_ref(dy, Q.quad_pack_ref(Q.flipped_transposed(w9), 2), 1)
    assert torch.allclose(dx, dxq, atol=1e-12)
    dwq = Q.quad_fold_ref(Q.wgradq_ref(x.detach(), dy, 0), 0)
    assert torch.allclose(dw, dwq, atol=1e-11)
    # UP: x low [N, H, W, C]
    x = torch.randn(N, H, W, C, generator=g, dtype=torch.float64).requires_grad_(True)
    y = Q.up_conv_torch(x, wt)
    yq = Q.convq_ref(x.detach(), Q.quad_pack_ref(w9, 1), 1)
    assert torch.allclose(y.detach(), yq, atol=1e-12)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    dx, dw = torch.autograd.grad(y, (x, wt), dy)
    dxq = Q.convq_ref(dy, Q.quad_pack_ref(Q.flipped_transposed(w9), 3), 0)
    assert torch.allclose(dx, dxq, atol=1e-12)
    dwq = Q.quad_fold_ref(Q.wgradq_ref(x.detach(), dy, 1), 1)
    assert torch.allclose(dw, dwq, atol=1e-11)


def test_quad_fold_is_transpose_of_pack():
    g = torch.Generator().manual_seed(6)
    for mode in (0, 1):
        w = torch.randn(3, 3, 3, 2, generator=g, dtype=torch.float64)
        q = torch.randn(3, 4, 4, 2, generator=g, dtype=torch.float64)
        assert abs(float((Q.quad_pack_ref(w, mode) * q).sum() - (w * Q.quad_fold_ref(q, mode)).sum())) < 1e-10


def test_oracle_quad_emulation_is_the_reference_op_without_rounding():
    """oracle/restate.py conv_pool_quad / conv_up_quad (the bf16-emulating oracle's restatement of the quad kernels' filter rounding) with the
    rounding switched off are exactly conv3x3 + avg_pool2d / interpolate + conv3x3, values and gradients"""
    import torch.nn.functional as F_
    from oracle import restate as O
    g = torch.Generator().manual_seed(9)
    w = torch.randn(64, 32, 3, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(64, generator=g, dtype=torch.float64)
    P = {"c.weight": w, "c.bias": b}
    assert O.quad_eligible(w)
    x = torch.randn(2, 32, 8, 8, generator=g, dtype=torch.float64).requires_grad_(True)
    for fn, ref in ((O.conv_pool_quad, lambda: F_.avg_pool2d(F_.conv2d(x, w, b, padding=1), 2)),
                    (O.conv_up_quad, lambda: F_.conv2d(F_.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1))):
        y, yr = fn(x, P, {}, "c", False, O.IDENT), ref()
        assert torch.allclose(y, yr, atol=1e-12)
        gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        ga = torch.autograd.grad(y, (x, w), gy)
        gb = torch.autograd.grad(yr, (x, w), gy)
        assert torch.allclose(ga[0], gb[0], atol=1e-11) and torch.allclose(ga[1], gb[1], atol=1e-10)
