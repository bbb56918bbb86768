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
