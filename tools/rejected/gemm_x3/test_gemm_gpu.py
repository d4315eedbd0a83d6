"""GPU: the heads' hand-written GEMM (csrc/gemm_x3.hip, augcl_gemm_f32 through head_ops.gemm) -- what nn.Linear and its backward
hand to the GEMM library in the reference (src/algorithms/modules.py:200-215, 247-261) -- against the same products in float64.
Tolerance (written here): |C - C64| <= 3e-7 * (|A| |B|) + 1e-30 element-wise, i.e. float32 rounding of the accumulated sum; the
library's float32 GEMM is held to the same yardstick in the same test so the bound is known to be a float32-level one."""
import pytest
import torch

from augcl_amd import head_ops

pytestmark = pytest.mark.gpu

TOL = 3e-7


def _ref(a, b, a_rows, b_rows, bias, relu):
    A, Bm = a.double(), b.double()
    A = A.transpose(-1, -2) if a_rows else A
    Bm = Bm if b_rows else Bm.transpose(-1, -2)
    c = A @ Bm
    mag = A.abs() @ Bm.abs()
    if bias is not None:
        bb = bias.double()
        c = c + (bb.unsqueeze(-2) if bb.dim() == 2 else bb)
        mag = mag + (bb.abs().unsqueeze(-2) if bb.dim() == 2 else bb.abs())
    return (torch.relu(c) if relu else c), mag


# (M, N, K, batch, a_rows, b_rows): every GEMM of a SVEA / SAC update at B = 256 (critic on 512 rows), A = 6, plus edge shapes
SHAPES = [
    (512, 2048, 106, 1, 0, 0), (512, 1024, 1024, 2, 0, 0), (256, 1024, 1024, 2, 0, 0),   # twin forward
    (512, 1024, 1024, 2, 0, 1), (512, 106, 2048, 1, 0, 1),                               # twin backward: dz1, dx
    (2048, 106, 512, 1, 1, 1), (1024, 1024, 512, 2, 1, 1),                               # twin dW1, dW2
    (256, 1024, 100, 1, 0, 0), (256, 12, 1024, 1, 0, 0), (256, 1024, 12, 1, 0, 1),       # actor mlp
    (12, 1024, 256, 1, 1, 1), (100, 14112, 512, 1, 1, 1),                                # actor dW3, projection dW
    (1, 1, 1, 1, 0, 0), (65, 63, 130, 3, 0, 0), (33, 70, 67, 1, 0, 1), (70, 33, 5, 2, 1, 1), (1, 1024, 1024, 1, 0, 0),
]


@pytest.mark.parametrize("M,N,K,batch,a_rows,b_rows", SHAPES)
def test_gemm_against_float64(cuda, M, N, K, batch, a_rows, b_rows):
    g = torch.Generator(device=cuda).manual_seed(M * 7 + N * 3 + K + batch)
    bs = (batch,) if batch > 1 else ()
    a = torch.randn(*bs, *((K, M) if a_rows else (M, K)), device=cuda, generator=g)
    b = torch.randn(*bs, *((K, N) if b_rows else (N, K)), device=cuda, generator=g) * 0.05
    a = a * torch.exp2(torch.randint(-12, 12, a.shape, device=cuda, generator=g).float())  # a wide dynamic range: no tensor-wide scale is assumed
    bias = torch.randn(*bs, N, device=cuda, generator=g) if (M + N) % 2 else None
    for relu in (False, True):
        want, mag = _ref(a, b, a_rows, b_rows, bias, relu)
        got = head_ops.gemm(a, b, a_rows=bool(a_rows), b_rows=bool(b_rows), bias=bias, relu=relu)
        assert got.shape == want.shape
        err = (got.double() - want).abs()
        assert bool((err <= TOL * mag + 1e-30).all()), float((err / (mag + 1e-30)).max())
        again = head_ops.gemm(a, b, a_rows=bool(a_rows), b_rows=bool(b_rows), bias=bias, relu=relu)
        assert torch.equal(got, again)  # fixed summation order, also through split K
    # the library's float32 GEMM on the same yardstick (so TOL is known to be float32-level, not a loosened bound)
    A32 = a.transpose(-1, -2) if a_rows else a
    B32 = b if b_rows else b.transpose(-1, -2)
    lib = A32 @ B32
    want, mag = _ref(a, b, a_rows, b_rows, None, False)
    lib_rel = float(((lib.double() - want).abs() / (mag + 1e-30)).max())
    got = head_ops.gemm(a, b, a_rows=bool(a_rows), b_rows=bool(b_rows))
    hip_rel = float(((got.double() - want).abs() / (mag + 1e-30)).max())
    assert hip_rel <= max(TOL, 2.0 * lib_rel), (hip_rel, lib_rel)


def test_gemm_strided_views_as_the_twin_trunks_use_them(cuda):
    """operands and results that are views: the (B, 2H) activation read as two (B, H) trunks, the result written into a column
    block, a 106-float leading dimension (not 16-byte aligned), a shared 2-D operand under a batched one"""
    g = torch.Generator(device=cuda).manual_seed(11)
    B, H, D = 96, 128, 106
    a1 = torch.randn(B, 2 * H, device=cuda, generator=g)
    W2 = torch.randn(2, H, H, device=cuda, generator=g) * 0.1
    z = head_ops.gemm(a1.view(B, 2, H).transpose(0, 1), W2)
    want = torch.bmm(a1.view(B, 2, H).transpose(0, 1).double(), W2.double().transpose(1, 2))
    assert float((z.double() - want).abs().max()) < 1e-4
    dz1 = torch.full((B, 2 * H), 7.0, device=cuda)
    head_ops.gemm(z, W2, b_rows=True, out=dz1.view(B, 2, H).transpose(0, 1))
    want = torch.bmm(z.double(), W2.double())
    assert float((dz1.view(B, 2, H).transpose(0, 1).double() - want).abs().max()) < 1e-3
    x = torch.randn(B, D, device=cuda, generator=g)
    W1 = torch.randn(2 * H, D, device=cuda, generator=g)
    assert float((head_ops.gemm(x, W1).double() - x.double() @ W1.double().t()).abs().max()) < 1e-4
    assert float((head_ops.gemm(dz1, W1, b_rows=True).double() - dz1.double() @ W1.double()).abs().max()) < 2e-2
    assert float((head_ops.gemm(dz1, x, a_rows=True, b_rows=True).double() - dz1.double().t() @ x.double()).abs().max()) < 2e-2
    shared = head_ops.gemm(x, W2.new_zeros(2, 5, D) + W1[:5])
    assert shared.shape == (2, B, 5) and torch.equal(shared[0], shared[1])


def test_gemm_split_k_tickets_reset_themselves(cuda):
    """two different split-K shapes, alternating, many times: the ticket words every launch leaves behind must be zero again"""
    g = torch.Generator(device=cuda).manual_seed(5)
    a1, b1 = torch.randn(64, 2048, device=cuda, generator=g), torch.randn(64, 2048, device=cuda, generator=g)
    a2, b2 = torch.randn(200, 1024, device=cuda, generator=g), torch.randn(40, 1024, device=cuda, generator=g)
    r1, r2 = head_ops.gemm(a1, b1), head_ops.gemm(a2, b2)
    for _ in range(20):
        assert torch.equal(head_ops.gemm(a1, b1), r1) and torch.equal(head_ops.gemm(a2, b2), r2)
    assert float((r1.double() - a1.double() @ b1.double().t()).abs().max()) < 1e-3
