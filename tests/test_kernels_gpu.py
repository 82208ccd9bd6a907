"""Per-kernel parity: each HIP kernel family through its C-ABI entry point (seld_k_*) against the
CPU oracle / a plain torch-CPU fp32-or-fp64 restatement of the same op.  Tolerance: 1e-4 relative
(tensor-normalised), the bar BASELINE.json's north_star states for fp32."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import REL_TOL, check, dev, ptr, rel_err

pytestmark = pytest.mark.gpu


def _conv_ref(x, w, b):
    xt = torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    wt = torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)
    return F.conv2d(xt, wt, torch.as_tensor(b, dtype=torch.float64), padding=1).permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("B,H,W,Cin", [(2, 50, 64, 7), (1, 7, 64, 7), (2, 30, 64, 10), (2, 20, 16, 64), (3, 10, 4, 64), (1, 37, 16, 64), (2, 45, 8, 64),
                                       (3, 70, 8, 64)])
def test_conv_fwd(seld_lib, B, H, W, Cin):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, 64)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    xd, wd, bd = dev(x), dev(w), dev(b)
    ref = _conv_ref(x, w, b)
    for mode in ((1, 0) if Cin == 64 else (1,)):     # Cin = 64: split-bf16 (default) and f32-MFMA kernels
        assert seld_lib.seld_k_set_option(b"conv64_split_bf16", mode) == 0
        z = torch.full((B, H, W, 64), float("nan"), device="cuda")
        st = torch.zeros(128, device="cuda")
        assert seld_lib.seld_k_conv3x3_fwd(ptr(xd), ptr(wd), ptr(bd), ptr(z), ptr(st), B, H, W, Cin, 64) == 0
        check(f"conv_fwd z {B,H,W,Cin} mode{mode}", z.cpu().numpy(), ref, tol=2e-6)
        s = st.cpu().numpy()
        check("conv_fwd sum(z)", s[:64], ref.sum(axis=(0, 1, 2)), tol=1e-4 * np.sqrt(ref.size / 64))
        check("conv_fwd sum(z^2)", s[64:], (ref ** 2).sum(axis=(0, 1, 2)))
    seld_lib.seld_k_set_option(b"conv64_split_bf16", 1)


@pytest.mark.parametrize("B,H,W", [(2, 20, 16), (3, 10, 4), (1, 37, 16), (4, 100, 16), (3, 70, 8)])
def test_conv_dgrad(seld_lib, B, H, W):
    rng = np.random.default_rng(2)
    dz = rng.standard_normal((B, H, W, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, 64)) / 24).astype(np.float32)
    x = torch.zeros((B, H, W, 64), dtype=torch.float64, requires_grad=True)
    wt = torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)
    y = F.conv2d(x.permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1)
    (g,) = torch.autograd.grad(y, x, torch.as_tensor(dz, dtype=torch.float64))
    dzd, wd = dev(dz), dev(w)
    # mode 1 = split-bf16: the six-product form (fp32-level, 2e-6) and the four-product form backward products take by default (option
    # "bwd_four_products": the two lo-factor terms dropped, a rounded mid plane; 2e-5 here, one decade inside the 1e-4 bar); mode 0 = the f32-input kernel
    try:
        for mode, four, tol in ((1, 0, 2e-6), (1, 1, 2e-5), (0, 0, 2e-6)):
            assert seld_lib.seld_k_set_option(b"conv64_split_bf16", mode) == 0
            assert seld_lib.seld_k_set_option(b"bwd_four_products", four) == 0
            dx = torch.full((B, H, W, 64), float("nan"), device="cuda")
            assert seld_lib.seld_k_conv3x3_dgrad(ptr(dzd), ptr(wd), ptr(dx), B, H, W, 64, 64) == 0
            check(f"conv_dgrad {B,H,W} mode{mode} four{four}", dx.cpu().numpy(), g.numpy(), tol=tol)
    finally:
        seld_lib.seld_k_set_option(b"conv64_split_bf16", 1)
        seld_lib.seld_k_set_option(b"bwd_four_products", 1)


@pytest.mark.parametrize("B,H,W,Cin", [(2, 50, 64, 7), (1, 7, 64, 7), (2, 30, 64, 10), (1, 9, 64, 10), (2, 20, 16, 64), (3, 10, 4, 64), (1, 37, 16, 64),
                                       (1, 3, 4, 64), (2, 33, 4, 64), (4, 600, 16, 64), (4, 600, 4, 64), (2, 33, 8, 64), (1, 5, 8, 64), (4, 600, 8, 64)])
def test_conv_wgrad(seld_lib, B, H, W, Cin):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    dz = rng.standard_normal((B, H, W, 64)).astype(np.float32)
    w = torch.zeros((3, 3, Cin, 64), dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(64, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), bias, padding=1).permute(0, 2, 3, 1)
    gw, gb = torch.autograd.grad(y, (w, bias), torch.as_tensor(dz, dtype=torch.float64))
    dw = torch.full((3, 3, Cin, 64), float("nan"), device="cuda")
    db = torch.full((64,), float("nan"), device="cuda")
    xd, dzd = dev(x), dev(dz)
    # 64 -> 64: split-bf16 with transposed LDS reads (conv_wgrad_sb.hip, default) and the f32-input MFMA kernel
    # (the split-bf16 kernel in both of its forms: six products, and the four the backward pass takes by default — option "bwd_four_products")
    for mode, four in (((1, 1), (1, 0), (0, 0)) if Cin == 64 else ((1, 1),)):
        dw.fill_(float("nan")); db.fill_(float("nan"))
        assert seld_lib.seld_k_set_option(b"conv64_split_bf16", mode) == 0
        assert seld_lib.seld_k_set_option(b"bwd_four_products", four) == 0
        try:
            assert seld_lib.seld_k_conv3x3_wgrad(ptr(xd), ptr(dzd), ptr(dw), ptr(db), B, H, W, Cin, 64) == 0
        finally:
            seld_lib.seld_k_set_option(b"conv64_split_bf16", 1)
            seld_lib.seld_k_set_option(b"bwd_four_products", 1)
        check(f"conv_wgrad dw {B,H,W,Cin} mode{mode} four{four}", dw.cpu().numpy(), gw.numpy())
        check(f"conv_wgrad db {B,H,W,Cin} mode{mode}", db.cpu().numpy(), gb.numpy())


@pytest.mark.parametrize("B,H,Cin", [(2, 50, 7), (1, 5, 7), (3, 35, 7), (2, 30, 10), (20, 300, 7)])
def test_conv_first_fwd_pool(seld_lib, B, H, Cin):
    """conv_pool.hip: the conv whose epilogue reduces the (5,4) pooling windows.  z and the statistics as the
    plain conv; zext against the window max/min of the kernel's OWN z (bit-exact: it is a selection), and
    bn_relu_ext(zext) bit-identical to the unfused bn_relu_pool_fwd(z) — including gamma < 0 and gamma = 0."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, H, 64, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, 64)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    gamma = (rng.uniform(0.5, 1.5, 64) * np.where(rng.random(64) < 0.3, -1, 1)).astype(np.float32)
    gamma[5] = 0.0
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(gamma)
    z = torch.full((B, H, 64, 64), float("nan"), device="cuda")
    ze = torch.full((B, H // 5, 16, 64), float("nan"), device="cuda")
    am = torch.full((B, H // 5, 16, 64), 255, device="cuda", dtype=torch.uint8)
    st = torch.zeros(128, device="cuda")
    assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), ptr(z), ptr(ze), ptr(am), ptr(st), B, H, Cin) == 0
    ref = _conv_ref(x, w, b)
    zh = z.cpu().numpy()
    check(f"conv_first_fwd_pool z {B,H,Cin}", zh, ref, tol=2e-6)
    s = st.cpu().numpy()
    check("conv_first_fwd_pool sum(z)", s[:64], ref.sum(axis=(0, 1, 2)), tol=1e-4 * np.sqrt(ref.size / 64))
    check("conv_first_fwd_pool sum(z^2)", s[64:], (ref ** 2).sum(axis=(0, 1, 2)))
    win = zh.reshape(B, H // 5, 5, 16, 4, 64)
    want = np.where(gamma < 0, win.min(axis=(2, 4)), win.max(axis=(2, 4)))
    np.testing.assert_array_equal(ze.cpu().numpy(), want)
    # amax: position row*4+col of that extreme in its window (random data: no ties)
    flat = np.where(gamma < 0, -win, win).transpose(0, 1, 3, 5, 2, 4).reshape(B, H // 5, 16, 64, 20)
    np.testing.assert_array_equal(am.cpu().numpy(), flat.argmax(-1).astype(np.uint8))
    # z not stored: inference (no amax either) and the z-free training path (amax only); z without amax is refused
    ze2 = torch.full_like(ze, float("nan"))
    am2 = torch.full_like(am, 255)
    assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), ptr(z), ptr(ze2), None, None, B, H, Cin) != 0
    # ... on the f32-input MFMA path: the same bits as the z-storing kernel
    assert seld_lib.seld_k_set_option(b"conv1_split_bf16", 0) == 0
    try:
        assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), None, ptr(ze2), ptr(am2), None, B, H, Cin) == 0
        assert torch.equal(ze, ze2) and torch.equal(am, am2)
        ze2.fill_(float("nan"))
        assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), None, ptr(ze2), None, None, B, H, Cin) == 0
        assert torch.equal(ze, ze2)
    finally:
        seld_lib.seld_k_set_option(b"conv1_split_bf16", 1)
    # ... on the default split-bf16 path (conv_pool_sb.hip): fp32-level values, so the window extreme is compared with
    # the float64 reference's, and the recorded position must hold a value within rounding of that extreme
    st2 = torch.zeros(128, device="cuda")
    for with_amax in (True, False):
        ze2.fill_(float("nan")); am2.fill_(255)
        assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), None, ptr(ze2), ptr(am2) if with_amax else None,
                                                   ptr(st2), B, H, Cin) == 0
        rwin = ref.reshape(B, H // 5, 5, 16, 4, 64)
        rwant = np.where(gamma < 0, rwin.min(axis=(2, 4)), rwin.max(axis=(2, 4)))
        check(f"conv_first_fwd_pool(split-bf16) zext {B,H,Cin}", ze2.cpu().numpy(), rwant, tol=2e-6)
        s2_ = st2.cpu().numpy()
        check("conv_first_fwd_pool(split-bf16) sum(z)", s2_[:64], ref.sum(axis=(0, 1, 2)), tol=1e-4 * np.sqrt(ref.size / 64))
        check("conv_first_fwd_pool(split-bf16) sum(z^2)", s2_[64:], (ref ** 2).sum(axis=(0, 1, 2)))
        if with_amax:
            rflat = np.where(gamma < 0, -rwin, rwin).transpose(0, 1, 3, 5, 2, 4).reshape(B, H // 5, 16, 64, 20)
            at = np.take_along_axis(rflat, am2.cpu().numpy().astype(np.int64)[..., None], axis=-1)[..., 0]
            assert am2.max().item() < 20
            assert np.abs(at - rflat.max(-1)).max() <= 2e-6 * np.abs(ref).max()
            assert (am2 == am).float().mean().item() > 0.999     # near-ties aside, the same positions as the fp32 kernel
    # pooled activation: elementwise over zext == BN+ReLU+MaxPool over z, bit for bit
    mean, var = zh.mean(axis=(0, 1, 2), dtype=np.float64), zh.var(axis=(0, 1, 2), dtype=np.float64)
    scale = (gamma / np.sqrt(var + 1e-3)).astype(np.float32)
    shift = (rng.normal(0, 0.3, 64) - mean * scale).astype(np.float32)
    sc, sh = dev(scale), dev(shift)
    p_ref = torch.full((B, H // 5, 16, 64), float("nan"), device="cuda")
    assert seld_lib.seld_k_bn_relu_pool_fwd(ptr(z), ptr(sc), ptr(sh), ptr(p_ref), B, H, 64, 64, 5, 4) == 0
    assert seld_lib.seld_k_bn_relu_ext(ptr(ze), ptr(sc), ptr(sh), ptr(ze), ze.numel()) == 0     # in place
    assert torch.equal(ze, p_ref)
    # H not a multiple of the pooling height is refused, not mis-tiled
    assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), None, ptr(ze2), None, None, B, H - 1, Cin) != 0


@pytest.mark.parametrize("B,H,W,pt,pf", [(2, 50, 64, 5, 4), (2, 10, 16, 1, 4), (3, 10, 4, 1, 2)])
def test_bn_relu_pool(seld_lib, B, H, W, pt, pf):
    rng = np.random.default_rng(4)
    z = rng.standard_normal((B, H, W, 64)).astype(np.float32)
    gamma = (rng.uniform(0.5, 1.5, 64) * np.where(rng.random(64) < 0.2, -1, 1)).astype(np.float32)
    beta = rng.normal(0, 0.3, 64).astype(np.float32)
    gamma[5], beta[5] = 0.0, 0.25    # degenerate channel: y constant > 0, argmax = first window element
    gamma[9], beta[9] = 0.0, -0.25   # degenerate channel killed by the ReLU
    zt = torch.as_tensor(z, dtype=torch.float64, ).requires_grad_(True)
    g = torch.as_tensor(gamma, dtype=torch.float64).requires_grad_(True)
    bt = torch.as_tensor(beta, dtype=torch.float64).requires_grad_(True)
    mean = zt.mean(dim=(0, 1, 2))
    var = ((zt - mean) ** 2).mean(dim=(0, 1, 2))
    invstd = torch.rsqrt(var + 1e-3)
    y = (zt - mean) * invstd * g + bt
    p = F.max_pool2d(torch.relu(y).permute(0, 3, 1, 2), (pt, pf), (pt, pf)).permute(0, 2, 3, 1)
    dp = rng.standard_normal(tuple(p.shape)).astype(np.float32)
    gz, gg, gb = torch.autograd.grad(p, (zt, g, bt), torch.as_tensor(dp, dtype=torch.float64))
    scale = (g * invstd).detach().numpy().astype(np.float32)
    shift = (bt - mean * g * invstd).detach().numpy().astype(np.float32)
    zd = dev(z)
    pd = torch.full(tuple(p.shape), float("nan"), device="cuda")
    sc, sh = dev(scale), dev(shift)
    assert seld_lib.seld_k_bn_relu_pool_fwd(ptr(zd), ptr(sc), ptr(sh), ptr(pd), B, H, W, 64, pt, pf) == 0
    check(f"bn_relu_pool_fwd {B,H,W,pt,pf}", pd.cpu().numpy(), p.detach().numpy())
    dz = torch.full((B, H, W, 64), float("nan"), device="cuda")
    dg = torch.full((64,), float("nan"), device="cuda")
    dbt = torch.full((64,), float("nan"), device="cuda")
    md, isd, gd, bd, dpd = dev(mean.detach().numpy()), dev(invstd.detach().numpy()), dev(gamma), dev(beta), dev(dp)
    assert seld_lib.seld_k_bn_relu_pool_bwd(ptr(zd), ptr(dpd), ptr(md), ptr(isd), ptr(gd), ptr(bd), ptr(dz), ptr(dg), ptr(dbt),
                                            B, H, W, 64, pt, pf) == 0
    check("bn_relu_pool_bwd dz", dz.cpu().numpy(), gz.numpy())
    check("bn_relu_pool_bwd dgamma", dg.cpu().numpy(), gg.numpy())
    check("bn_relu_pool_bwd dbeta", dbt.cpu().numpy(), gb.numpy())


@pytest.mark.parametrize("M,N,K,transb,act", [(200, 384, 128, 0, 0), (130, 12, 128, 0, 1), (77, 36, 128, 0, 2),
                                              (200, 128, 384, 1, 0), (99, 128, 12, 1, 0), (64, 128, 36, 1, 0), (70, 50, 30, 0, 0)])
def test_gemm(seld_lib, M, N, K, transb, act):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = (rng.standard_normal((N, K) if transb else (K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ (Bm.T if transb else Bm).astype(np.float64) + bias
    if act == 1:
        ref = 1 / (1 + np.exp(-ref))
    elif act == 2:
        ref = np.tanh(ref)
    Cd = torch.full((M, N), float("nan"), device="cuda")
    Ad, Bd, bd = dev(A), dev(Bm), dev(bias)
    assert seld_lib.seld_k_gemm(ptr(Ad), ptr(Bd), ptr(bd), ptr(Cd), M, N, K, transb, act, 0) == 0
    check(f"gemm {M,N,K,transb,act}", Cd.cpu().numpy(), ref)
    # accumulate form
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    Cd = dev(C0)
    assert seld_lib.seld_k_gemm(ptr(Ad), ptr(Bd), None, ptr(Cd), M, N, K, transb, 0, 1) == 0
    check("gemm accumulate", Cd.cpu().numpy(), C0 + A.astype(np.float64) @ (Bm.T if transb else Bm).astype(np.float64))


@pytest.mark.parametrize("M,N,K,transb", [(1200, 384, 128, 0), (333, 384, 64, 0), (200, 128, 128, 0), (1200, 128, 384, 1),
                                          (77, 128, 128, 1), (50, 30, 64, 0)])
def test_gemm_pairs(seld_lib, M, N, K, transb):
    """Merged launches: two products sharing A, and one product over a concatenated K axis."""
    rng = np.random.default_rng(15)
    A0, A1 = (rng.standard_normal((M, K)).astype(np.float32) for _ in range(2))
    B0, B1 = ((rng.standard_normal((N, K) if transb else (K, N)) / np.sqrt(K)).astype(np.float32) for _ in range(2))
    b0, b1 = (rng.standard_normal(N).astype(np.float32) for _ in range(2))
    op = (lambda m: m.T.astype(np.float64)) if transb else (lambda m: m.astype(np.float64))
    C0 = torch.full((M, N), float("nan"), device="cuda")
    C1 = torch.full((M, N), float("nan"), device="cuda")
    A0d, A1d, B0d, B1d, b0d, b1d = dev(A0), dev(A1), dev(B0), dev(B1), dev(b0), dev(b1)
    assert seld_lib.seld_k_gemm_pair_n(ptr(A0d), ptr(B0d), ptr(B1d), ptr(b0d), ptr(b1d), ptr(C0), ptr(C1), M, N, K, transb, 0) == 0
    check("pair_n first", C0.cpu().numpy(), A0.astype(np.float64) @ op(B0) + b0)
    check("pair_n second", C1.cpu().numpy(), A0.astype(np.float64) @ op(B1) + b1)
    # the merged launch returns the very bits of two separate launches
    S = torch.empty((M, N), device="cuda")
    assert seld_lib.seld_k_gemm(ptr(A0d), ptr(B1d), ptr(b1d), ptr(S), M, N, K, transb, 0, 0) == 0
    assert torch.equal(S, C1)
    C = torch.full((M, N), float("nan"), device="cuda")
    rc = seld_lib.seld_k_gemm_pair_k(ptr(A0d), ptr(A1d), ptr(B0d), ptr(B1d), None, ptr(C), M, N, K, transb, 0, 0)
    assert rc == 0
    check("pair_k", C.cpu().numpy(), A0.astype(np.float64) @ op(B0) + A1.astype(np.float64) @ op(B1))


def test_gemm_pair_k_refuses_ragged_k(seld_lib):
    t = torch.zeros(64 * 64, device="cuda")
    assert seld_lib.seld_k_gemm_pair_k(ptr(t), ptr(t), ptr(t), ptr(t), None, ptr(t), 8, 8, 20, 0, 0, 0) != 0


@pytest.mark.parametrize("M,N,K,transb,mode", [(1200, 384, 128, 0, 1), (333, 128, 128, 0, 1), (200, 128, 64, 0, 0),
                                               (1200, 128, 384, 1, 2), (77, 128, 128, 1, 2), (130, 256, 96, 1, 0),
                                               (19200, 384, 128, 0, 1), (19200, 128, 384, 1, 2)])
def test_gemm_split_bf16(seld_lib, M, N, K, transb, mode):
    """Split-bf16 products (gemm_sb.hip) against float64, with operands spanning many binades so that the mid / lo
    terms matter; the last two cases are the model's shapes at BASELINE.json's batch (32 clips x 600 frames)."""
    rng = np.random.default_rng(16)
    scale = lambda shape: np.exp2(rng.integers(-6, 7, size=shape)).astype(np.float32)
    A0, A1 = ((rng.standard_normal((M, K)) * scale((M, K))).astype(np.float32) for _ in range(2))
    bshape = (N, K) if transb else (K, N)
    B0, B1 = ((rng.standard_normal(bshape) * scale(bshape) / np.sqrt(K)).astype(np.float32) for _ in range(2))
    b0, b1 = (rng.standard_normal(N).astype(np.float32) for _ in range(2))
    op = (lambda m: m.T.astype(np.float64)) if transb else (lambda m: m.astype(np.float64))
    C0 = torch.full((M, N), float("nan"), device="cuda")
    C1 = torch.full((M, N), float("nan"), device="cuda")
    A0d, A1d, B0d, B1d, b0d, b1d = dev(A0), dev(A1), dev(B0), dev(B1), dev(b0), dev(b1)
    rc = seld_lib.seld_k_gemm_sb(ptr(A0d), ptr(A1d), ptr(B0d), ptr(B1d), ptr(b0d), ptr(b1d), ptr(C0), ptr(C1), M, N, K, transb, 0,
                                 mode)
    assert rc == 0
    ref0 = A0.astype(np.float64) @ op(B0) + b0
    if mode == 2:
        ref0 = ref0 + A1.astype(np.float64) @ op(B1)
    # |error| against the product of magnitudes (what an fp32 dot product of these operands can promise)
    mag = np.abs(A0).astype(np.float64) @ np.abs(op(B0)) + (np.abs(A1).astype(np.float64) @ np.abs(op(B1)) if mode == 2 else 0) + 1.0
    err = np.abs(C0.cpu().numpy() - ref0) / mag
    assert err.max() < 2e-6, err.max()
    check("gemm_sb", C0.cpu().numpy(), ref0)
    if mode == 1:
        check("gemm_sb second product", C1.cpu().numpy(), A0.astype(np.float64) @ op(B1) + b1)


@pytest.mark.parametrize("M,N,mode", [(19200, 384, 1), (1203, 256, 0), (33, 384, 1), (4000, 128, 1)])
def test_gemm_k128_b_stationary_form_is_bit_identical(seld_lib, M, N, mode):
    """The opt-in K = 128 form with B stationary in LDS (gemm_sbp_kernel: persistent workgroups, no barrier after the prologue; option gsb_dbg
    bit 6, round 5) takes the products and k-steps in the tiled kernel's order: the same bits, ragged row counts included."""
    rng = np.random.default_rng(M + N)
    A = dev((rng.standard_normal((M, 128)) * np.exp2(rng.integers(-6, 7, size=(M, 128)))).astype(np.float32))
    B0, B1 = (dev(rng.standard_normal((128, N)).astype(np.float32) / 11.0) for _ in range(2))
    b0, b1 = (dev(rng.standard_normal(N).astype(np.float32)) for _ in range(2))
    out = []
    for dbg in (0, 64):
        C0, C1 = (torch.full((M, N), float("nan"), device="cuda") for _ in range(2))
        assert seld_lib.seld_k_set_option(b"gsb_dbg", dbg) == 0
        try:
            assert seld_lib.seld_k_gemm_sb(ptr(A), None, ptr(B0), ptr(B1) if mode else None, ptr(b0), ptr(b1) if mode else None, ptr(C0),
                                           ptr(C1) if mode else None, M, N, 128, 0, 0, mode) == 0
        finally:
            seld_lib.seld_k_set_option(b"gsb_dbg", 0)
        out.append((C0.cpu().numpy(), C1.cpu().numpy()))
    assert np.isfinite(out[1][0]).all()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    if mode:
        np.testing.assert_array_equal(out[0][1], out[1][1])


def test_gemm_split_bf16_refuses_unsupported_shapes(seld_lib):
    t = torch.zeros(256 * 256, device="cuda")
    for N, K in ((100, 64), (128, 40)):
        assert seld_lib.seld_k_gemm_sb(ptr(t), None, ptr(t), None, None, None, ptr(t), None, 64, N, K, 0, 0, 0) != 0


@pytest.mark.parametrize("M,K1,N", [(1200, 128, 384), (333, 128, 12), (100, 128, 36), (40, 30, 50), (37, 128, 128), (19200, 128, 384)])
def test_gemm_tn(seld_lib, M, K1, N):
    """Weight-gradient products A^T B (+ column sums).  K1 = 128 with N % 128 == 0 runs on the split-bf16 kernel with transposed
    LDS reads (gemm_tn_sb.hip) by default; the f32-input kernel is checked on the same inputs."""
    rng = np.random.default_rng(6)
    A = (rng.standard_normal((M, K1)) * np.exp2(rng.integers(-5, 6, size=(M, K1)))).astype(np.float32)
    Bm = (rng.standard_normal((M, N)) * np.exp2(rng.integers(-5, 6, size=(M, N)))).astype(np.float32)
    Cd = torch.full((K1, N), float("nan"), device="cuda")
    cs = torch.full((N,), float("nan"), device="cuda")
    Ad, Bd = dev(A), dev(Bm)
    ref = A.astype(np.float64).T @ Bm.astype(np.float64)
    for mode, four in ((1, 1), (1, 0), (0, 0)):      # four: the split-bf16 kernel's four-product form (option "bwd_four_products", the default of a kernel gradient)
        assert seld_lib.seld_k_set_option(b"gemm_tn_split_bf16", mode) == 0
        assert seld_lib.seld_k_set_option(b"bwd_four_products", four) == 0
        try:
            Cd.fill_(float("nan")); cs.fill_(float("nan"))
            assert seld_lib.seld_k_gemm_tn(ptr(Ad), ptr(Bd), ptr(Cd), ptr(cs), M, K1, N) == 0
            check(f"gemm_tn {M,K1,N} mode{mode} four{four}", Cd.cpu().numpy(), ref)
            check(f"gemm_tn colsum {M,K1,N} mode{mode}", cs.cpu().numpy(), Bm.astype(np.float64).sum(0))
            Cd.fill_(float("nan"))
            assert seld_lib.seld_k_gemm_tn(ptr(Ad), ptr(Bd), ptr(Cd), None, M, K1, N) == 0
            check(f"gemm_tn (no colsum) {M,K1,N} mode{mode}", Cd.cpu().numpy(), ref)
        finally:
            seld_lib.seld_k_set_option(b"gemm_tn_split_bf16", 1)
            seld_lib.seld_k_set_option(b"bwd_four_products", 1)


def _gru_ref(gx, U, brec, reverse):
    """Keras GRU(reset_after=True) recurrence in fp64 torch with autograd (oracle twin)."""
    B, S, _ = gx.shape
    h = gx.new_zeros(B, 128)
    outs = [None] * S
    for t in (range(S - 1, -1, -1) if reverse else range(S)):
        gh = h @ U + brec
        z = torch.sigmoid(gx[:, t, :128] + gh[:, :128])
        r = torch.sigmoid(gx[:, t, 128:256] + gh[:, 128:256])
        hh = torch.tanh(gx[:, t, 256:] + r * gh[:, 256:])
        h = z * h + (1 - z) * hh
        outs[t] = h
    return torch.stack(outs, 1)


@pytest.mark.parametrize("B,S", [(2, 10), (3, 60), (1, 1)])
def test_gru_fwd_bwd(seld_lib, B, S):
    rng = np.random.default_rng(7)
    mk = lambda *s, sc=1.0: (rng.standard_normal(s) * sc).astype(np.float32)
    gx = [mk(B, S, 384), mk(B, S, 384)]
    U = [mk(128, 384, sc=1 / np.sqrt(128)), mk(128, 384, sc=1 / np.sqrt(128))]
    br = [mk(384, sc=0.1), mk(384, sc=0.1)]
    dout = mk(B, S, 128)
    tg = [torch.as_tensor(a, dtype=torch.float64).requires_grad_(True) for a in gx]
    tU = [torch.as_tensor(a, dtype=torch.float64).requires_grad_(True) for a in U]
    tb = [torch.as_tensor(a, dtype=torch.float64).requires_grad_(True) for a in br]
    hf = _gru_ref(tg[0], tU[0], tb[0], False)
    hb = _gru_ref(tg[1], tU[1], tb[1], True)
    out = hf * hb
    grads = torch.autograd.grad(out, tg + tU + tb, torch.as_tensor(dout, dtype=torch.float64))
    d = {k: dev(v) for k, v in dict(gxf=gx[0], gxb=gx[1], Uf=U[0], Ub=U[1], bf=br[0], bb=br[1], dout=dout).items()}
    nan = lambda *s: torch.full(s, float("nan"), device="cuda")
    h_f, h_b, o = nan(B, S, 128), nan(B, S, 128), nan(B, S, 128)
    sv_f, sv_b = nan(B, S, 4, 128), nan(B, S, 4, 128)
    assert seld_lib.seld_k_gru_fwd(ptr(d["gxf"]), ptr(d["gxb"]), ptr(d["Uf"]), ptr(d["Ub"]), ptr(d["bf"]), ptr(d["bb"]),
                                   ptr(h_f), ptr(h_b), ptr(sv_f), ptr(sv_b), ptr(o), B, S, 128) == 0
    check(f"gru_fwd h_f {B,S}", h_f.cpu().numpy(), hf.detach().numpy())
    check(f"gru_fwd h_b {B,S}", h_b.cpu().numpy(), hb.detach().numpy())
    check("gru_fwd out (mul merge)", o.cpu().numpy(), out.detach().numpy())
    dgx_f, dgx_b, dgh_f, dgh_b = nan(B, S, 384), nan(B, S, 384), nan(B, S, 384), nan(B, S, 384)
    assert seld_lib.seld_k_gru_bwd(ptr(d["dout"]), ptr(h_f), ptr(h_b), ptr(sv_f), ptr(sv_b), ptr(d["Uf"]), ptr(d["Ub"]),
                                   ptr(dgx_f), ptr(dgx_b), ptr(dgh_f), ptr(dgh_b), B, S, 128) == 0
    check("gru_bwd dgx_f", dgx_f.cpu().numpy(), grads[0].numpy())
    check("gru_bwd dgx_b", dgx_b.cpu().numpy(), grads[1].numpy())
    # recurrent kernel / recurrent bias gradients from dgh (what the TN GEMM + colsum compute)
    for dirn, (dgh, H, gU, gb) in enumerate(((dgh_f, hf, grads[2], grads[4]), (dgh_b, hb, grads[3], grads[5]))):
        g = dgh.cpu().numpy().astype(np.float64)
        Hn = H.detach().numpy()
        Hprev = np.zeros_like(Hn)
        if dirn == 0:
            Hprev[:, 1:] = Hn[:, :-1]
        else:
            Hprev[:, :-1] = Hn[:, 1:]
        check(f"gru_bwd dU dir{dirn}", np.einsum("bsk,bsn->kn", Hprev, g), gU.numpy())
        check(f"gru_bwd dbrec dir{dirn}", g.sum(axis=(0, 1)), gb.numpy())


@pytest.mark.parametrize("mode", ["MSE", "MMSE", "MAE", "MSLE"])
def test_losses(seld_lib, mode):
    import sys, os
    from seld_amd import _lib
    from oracle import seldnet_oracle as O
    rng = np.random.default_rng(8)
    B, S, nc = 3, 20, 12
    _, ys, yd = O.synthetic_batch(B, S * 5, seed=11)
    sp = rng.standard_normal((B, S, nc)) * 3
    sp[0, 0, :4] = [40, -40, 20, -20]  # saturated logits: exercises the BCE clip
    dpre = rng.standard_normal((B, S, 3 * nc))
    tsp = torch.as_tensor(sp, dtype=torch.float64).requires_grad_(True)
    tdp = torch.as_tensor(dpre, dtype=torch.float64).requires_grad_(True)
    sed, doa = torch.sigmoid(tsp), torch.tanh(tdp)
    lw = (1.0, 1000.0)
    obj, sl, dl = O.losses_and_objective(sed, doa, torch.as_tensor(ys, dtype=torch.float64), torch.as_tensor(yd, dtype=torch.float64), mode, lw)
    gs, gd = torch.autograd.grad(obj, (tsp, tdp))
    cfg = _lib.LossCfg({"MSE": 0, "MMSE": 1, "MAE": 2, "MSLE": 3}[mode], lw[0], lw[1], 1.0, 0.0)
    sedd, doad = dev(sed.detach().numpy()), dev(doa.detach().numpy())
    ysd, ydd = dev(ys), dev(yd)
    sl_d = torch.full((1,), float("nan"), device="cuda")
    dl_d = torch.full((B * S if mode != "MMSE" else 1,), float("nan"), device="cuda")
    gsd, gdd = torch.full((B, S, nc), float("nan"), device="cuda"), torch.full((B, S, 3 * nc), float("nan"), device="cuda")
    assert seld_lib.seld_k_losses(ptr(sedd), ptr(doad), ptr(ysd), ptr(ydd), C.byref(cfg), ptr(sl_d), ptr(dl_d), ptr(gsd), ptr(gdd), B, S, nc) == 0
    check(f"losses sloss {mode}", sl_d.cpu().numpy(), sl.detach().numpy().reshape(1))
    check(f"losses dloss {mode}", dl_d.cpu().numpy(), dl.detach().numpy().reshape(-1))
    # sigmoid computed in fp32 upstream: compare gradients where the fp32 sigmoid is not saturated
    check(f"losses dsed_pre {mode}", gsd.cpu().numpy(), gs.numpy(), tol=2e-4)
    check(f"losses ddoa_pre {mode}", gdd.cpu().numpy(), gd.numpy())


def test_adam(seld_lib):
    rng = np.random.default_rng(9)
    n = 10007
    th, g, m, v = (rng.standard_normal(n).astype(np.float32) for _ in range(4))
    v = np.abs(v)
    from oracle import seldnet_oracle as O
    rt, rm, rv = O.adam_update(torch.as_tensor(th, dtype=torch.float64), torch.as_tensor(g, dtype=torch.float64),
                               torch.as_tensor(m, dtype=torch.float64), torch.as_tensor(v, dtype=torch.float64), step=3, lr=1e-3)
    td, gd, md, vd = dev(th), dev(g), dev(m), dev(v)
    assert seld_lib.seld_k_adam(ptr(td), ptr(gd), ptr(md), ptr(vd), n, 1e-3, 0.9, 0.999, 1e-7, 3) == 0
    check("adam theta", td.cpu().numpy(), rt.numpy(), tol=1e-6)
    check("adam m", md.cpu().numpy(), rm.numpy(), tol=1e-6)
    check("adam v", vd.cpu().numpy(), rv.numpy(), tol=1e-6)


@pytest.mark.parametrize("B,H,CIN", [(2, 50, 7), (1, 20, 7), (2, 15, 7), (2, 25, 10)])
def test_conv1_bwd_fused(seld_lib, B, H, CIN):
    """conv1 + BN(training) + ReLU + MaxPool(5,4) backward in one fused pass vs autograd (fp64)."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((B, H, 64, CIN)).astype(np.float32)
    w = (rng.standard_normal((3, 3, CIN, 64)) / np.sqrt(9 * CIN)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32) * 0.1
    gamma = (rng.uniform(0.5, 1.5, 64) * np.where(rng.random(64) < 0.2, -1, 1)).astype(np.float32)
    beta = rng.normal(0, 0.3, 64).astype(np.float32)
    # forward z on the GPU (so that y == p is evaluated on the same fp32 z the product would see)
    xd, wd, bd = dev(x), dev(w), dev(b)
    zd = torch.empty((B, H, 64, 64), device="cuda")
    assert seld_lib.seld_k_conv3x3_fwd(ptr(xd), ptr(wd), ptr(bd), ptr(zd), None, B, H, 64, CIN, 64) == 0
    z = zd.cpu().numpy()
    # reference: autograd through conv -> BN(batch stats) -> relu -> pool in fp64, from the same fp32 inputs
    tw = torch.as_tensor(w, dtype=torch.float64).requires_grad_(True)
    tb = torch.as_tensor(b, dtype=torch.float64).requires_grad_(True)
    tg = torch.as_tensor(gamma, dtype=torch.float64).requires_grad_(True)
    tbe = torch.as_tensor(beta, dtype=torch.float64).requires_grad_(True)
    zt = F.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2), tw.permute(3, 2, 0, 1), tb, padding=1).permute(0, 2, 3, 1)
    mean = zt.mean(dim=(0, 1, 2))
    var = ((zt - mean) ** 2).mean(dim=(0, 1, 2))
    invstd = torch.rsqrt(var + 1e-3)
    y = (zt - mean) * invstd * tg + tbe
    p = F.max_pool2d(torch.relu(y).permute(0, 3, 1, 2), (5, 4), (5, 4)).permute(0, 2, 3, 1)
    dp = rng.standard_normal(tuple(p.shape)).astype(np.float32)
    gw, gb, gg, gbe = torch.autograd.grad(p, (tw, tb, tg, tbe), torch.as_tensor(dp, dtype=torch.float64))
    nan = lambda *s: torch.full(s, float("nan"), device="cuda")
    dw, db, dg, dbe = nan(3, 3, CIN, 64), nan(64), nan(64), nan(64)
    md, isd = dev(mean.detach().numpy()), dev(invstd.detach().numpy())
    gd, bed, dpd = dev(gamma), dev(beta), dev(dp)
    assert seld_lib.seld_k_conv1_bwd_fused(ptr(xd), ptr(zd), ptr(dpd), ptr(md), ptr(isd), ptr(gd), ptr(bed), ptr(dw), ptr(db),
                                           ptr(dg), ptr(dbe), B, H, CIN, 5, 4) == 0
    check(f"conv1_bwd_fused dw {B,H}", dw.cpu().numpy(), gw.numpy())
    check(f"conv1_bwd_fused dgamma {B,H}", dg.cpu().numpy(), gg.numpy())
    check(f"conv1_bwd_fused dbeta {B,H}", dbe.cpu().numpy(), gbe.numpy())
    assert np.abs(db.cpu().numpy()).max() <= 1e-3 * np.abs(gw.numpy()).max()   # exact-arithmetic zero (bias before BN)


def test_conv1_bwd_single_routing_on_ties(seld_lib):
    """MaxPoolGrad routes a window's gradient to ONE element.  With all-zero weights every z of a channel equals
    its bias — every window is a 20-way tie — so the kernel+bias gradient must equal the gradient of routing each
    dp to one position per window (the first in the kernel's scan order: position 0), not 20 copies of it."""
    B, H, CIN = 1, 10, 7
    rng = np.random.default_rng(21)
    x = rng.standard_normal((B, H, 64, CIN)).astype(np.float32)
    w = np.zeros((3, 3, CIN, 64), np.float32)
    b = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    gamma, beta = np.ones(64, np.float32), np.full(64, 0.25, np.float32)
    xd, wd, bd, gd, bed = dev(x), dev(w), dev(b), dev(gamma), dev(beta)
    zd = torch.empty((B, H, 64, 64), device="cuda")
    ze = torch.empty((B, H // 5, 16, 64), device="cuda")
    am = torch.full((B, H // 5, 16, 64), 255, device="cuda", dtype=torch.uint8)
    assert seld_lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), ptr(zd), ptr(ze), ptr(am), None, B, H, CIN) == 0
    assert (am == 0).all()                                        # ties: the first position of the scan
    # z is constant per channel -> xhat = 0, y = beta > 0, p = beta; batch mean = bias, variance = 0
    mean, invstd = b.copy(), np.full(64, 1.0 / np.sqrt(1e-3), np.float32)
    dp = rng.standard_normal((B, H // 5, 16, 64)).astype(np.float32)
    nan = lambda *s_: torch.full(s_, float("nan"), device="cuda")
    dw, db, dg, dbe = nan(3, 3, CIN, 64), nan(64), nan(64), nan(64)
    md, isd, dpd = dev(mean), dev(invstd), dev(dp)
    assert seld_lib.seld_k_conv1_bwd_fused(ptr(xd), ptr(zd), ptr(dpd), ptr(md), ptr(isd), ptr(gd), ptr(bed), ptr(dw), ptr(db),
                                           ptr(dg), ptr(dbe), B, H, CIN, 5, 4) == 0
    # reference: dy = dp at position 0 of each window (image row 5*tp, bin 4*fp), 0 elsewhere; dz = scale*(dy - mean(dy))
    # (xhat = 0 kills the second BN term); dW = sum_px patch(px)^T dz(px)
    scale = gamma.astype(np.float64) * invstd.astype(np.float64)
    dy = np.zeros((B, H, 64, 64))
    dy[:, ::5, ::4, :] = dp
    dz = scale * (dy - dy.mean(axis=(0, 1, 2)))
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.zeros((3, 3, CIN, 64))
    for kh in range(3):
        for kw in range(3):
            ref[kh, kw] = np.einsum("bhwc,bhwo->co", xp[:, kh:kh + H, kw:kw + 64, :], dz)
    check("single routing dw", dw.cpu().numpy(), ref)
    check("single routing dbeta", dbe.cpu().numpy(), dp.sum(axis=(0, 1, 2)))


def _patches(x):
    """im2col of a 3x3 'same' conv: [B*H*W, 9*Cin + 1] in fp64, k = (kh, kw, ci), last column = 1 (bias)."""
    B, H, W, Cin = x.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    cols = [xp[:, kh:kh + H, kw:kw + W, :] for kh in range(3) for kw in range(3)]
    P = np.concatenate(cols + [np.ones((B, H, W, 1))], axis=-1)
    return P.reshape(-1, 9 * Cin + 1)


@pytest.mark.parametrize("B,H,Cin", [(2, 50, 7), (1, 7, 7), (2, 10, 10), (3, 700, 7)])
def test_conv1_gram_matrix(seld_lib, B, H, Cin):
    """conv_gram.hip: G = P^T P of the zero-padded input patches (+ ones column), upper 32x32 tiles."""
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((B, H, 64, Cin)) + 0.3).astype(np.float32)
    KP = 64 if Cin == 7 else 128
    G = torch.full((KP, KP), float("nan"), device="cuda")
    xd = dev(x)
    assert seld_lib.seld_k_conv1_gram(ptr(xd), ptr(G), B, H, Cin) == 0
    P = _patches(x)
    ref = P.T @ P
    K1 = 9 * Cin + 1
    got = G.cpu().numpy()
    for r0 in range(0, KP, 32):
        for c0 in range(r0, KP, 32):                       # upper tiles only
            r1, c1 = min(r0 + 32, K1), min(c0 + 32, K1)
            if r0 < K1 and c0 < K1:
                check(f"gram tile {r0},{c0}", got[r0:r1, c0:c1], ref[r0:r1, c0:c1], tol=2e-6)
    assert got[K1 - 1, K1 - 1] == B * H * 64


@pytest.mark.parametrize("B,H,CIN", [(2, 50, 7), (1, 20, 7), (2, 15, 7), (2, 25, 10), (4, 600, 7)])
def test_conv1_train_without_pre_bn_tensor(seld_lib, B, H, CIN):
    """The first block trained from x alone (no z in memory): forward p and the four gradients against fp64 autograd
    through conv -> BN(batch statistics) -> ReLU -> MaxPool(5,4)."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((B, H, 64, CIN)).astype(np.float32)
    w = (rng.standard_normal((3, 3, CIN, 64)) / np.sqrt(9 * CIN)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32) * 0.1
    gamma = (rng.uniform(0.5, 1.5, 64) * np.where(rng.random(64) < 0.2, -1, 1)).astype(np.float32)
    beta = rng.normal(0, 0.3, 64).astype(np.float32)
    tw = torch.as_tensor(w, dtype=torch.float64).requires_grad_(True)
    tb = torch.as_tensor(b, dtype=torch.float64).requires_grad_(True)
    tg = torch.as_tensor(gamma, dtype=torch.float64).requires_grad_(True)
    tbe = torch.as_tensor(beta, dtype=torch.float64).requires_grad_(True)
    zt = F.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2), tw.permute(3, 2, 0, 1), tb, padding=1).permute(0, 2, 3, 1)
    mean = zt.mean(dim=(0, 1, 2))
    var = ((zt - mean) ** 2).mean(dim=(0, 1, 2))
    y = (zt - mean) * torch.rsqrt(var + 1e-3) * tg + tbe
    p = F.max_pool2d(torch.relu(y).permute(0, 3, 1, 2), (5, 4), (5, 4)).permute(0, 2, 3, 1)
    dp = rng.standard_normal(tuple(p.shape)).astype(np.float32)
    gw, gb, gg, gbe = torch.autograd.grad(p, (tw, tb, tg, tbe), torch.as_tensor(dp, dtype=torch.float64))
    nan = lambda *s_: torch.full(s_, float("nan"), device="cuda")
    xd, wd, bd, gd, bed, dpd = dev(x), dev(w), dev(b), dev(gamma), dev(beta), dev(dp)
    pd, dw, db, dg, dbe = nan(*p.shape), nan(3, 3, CIN, 64), nan(64), nan(64), nan(64)
    assert seld_lib.seld_k_conv1_train_gram(ptr(xd), ptr(wd), ptr(bd), ptr(gd), ptr(bed), ptr(dpd), ptr(pd), ptr(dw), ptr(db),
                                            ptr(dg), ptr(dbe), B, H, CIN) == 0
    check(f"z-free p {B,H}", pd.cpu().numpy(), p.detach().numpy(), tol=2e-6)
    check(f"z-free dw {B,H}", dw.cpu().numpy(), gw.numpy())
    check(f"z-free dgamma {B,H}", dg.cpu().numpy(), gg.numpy())
    check(f"z-free dbeta {B,H}", dbe.cpu().numpy(), gbe.numpy())
    assert np.abs(db.cpu().numpy()).max() <= 1e-3 * np.abs(gw.numpy()).max()   # exact-arithmetic zero (bias before BN)


# ---- resnet50_block pieces (spec/RESNET50_BLOCK.md) at the shapes its stages run them at: B*S*W pixels with W = 16, 8, 4, 2
@pytest.mark.parametrize("B,H,W,Cin,Cout,ksize,stride_f", [(2, 30, 16, 64, 32, 1, 1), (2, 30, 16, 32, 32, 3, 1), (2, 30, 16, 128, 64, 1, 2),
                                                           (4, 60, 2, 256, 256, 3, 1), (4, 60, 2, 256, 1024, 1, 1), (4, 60, 4, 512, 1024, 1, 2),
                                                           (3, 7, 4, 128, 128, 3, 1), (2, 30, 16, 32, 128, 1, 1), (2, 61, 4, 128, 512, 1, 1), (2, 21, 4, 128, 256, 3, 1), (5, 9, 2, 256, 128, 3, 1)])
@pytest.mark.parametrize("split", [1, 0])
def test_rn_conv_fwd_bwd(seld_lib, B, H, W, Cin, Cout, ksize, stride_f, split):
    """A resnet50_block convolution as its three products (forward, input gradient, kernel gradient) against float64 autograd: on the
    split-bf16 kernels where the shape allows (Cout, resp. K, a multiple of 128: the last five shapes at least in part) and on the fp32
    MFMA GEMM (`rn_split_bf16 = 0`, and every other shape)."""
    assert seld_lib.seld_k_set_option(b"rn_split_bf16", split) == 0
    try:
        _rn_conv_fwd_bwd(seld_lib, B, H, W, Cin, Cout, ksize, stride_f)
    finally:
        seld_lib.seld_k_set_option(b"rn_split_bf16", 1)


def _rn_conv_fwd_bwd(seld_lib, B, H, W, Cin, Cout, ksize, stride_f):
    rng = np.random.default_rng(31)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((ksize, ksize, Cin, Cout)) / np.sqrt(ksize * ksize * Cin)).astype(np.float32)
    Wo = W // stride_f
    dz = rng.standard_normal((B, H, Wo, Cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1).requires_grad_(True)
    zt = torch.nn.functional.conv2d(xt, wt, stride=(1, stride_f), padding=ksize // 2)
    zt.backward(torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2))
    z = torch.full((B, H, Wo, Cout), float("nan"), device="cuda")
    xd, wd, dzd = dev(x), dev(w), dev(dz)
    assert seld_lib.seld_k_rn_conv(ptr(xd), ptr(wd), ptr(z), B, H, W, Cin, Cout, ksize, stride_f) == 0
    check("rn_conv z", z.cpu().numpy(), zt.detach().permute(0, 2, 3, 1).numpy())
    dw = torch.full((ksize, ksize, Cin, Cout), float("nan"), device="cuda")
    dx = torch.full((B, H, W, Cin), float("nan"), device="cuda")
    assert seld_lib.seld_k_rn_conv_bwd(ptr(xd), ptr(wd), ptr(dzd), ptr(dw), ptr(dx), B, H, W, Cin, Cout, ksize, stride_f) == 0
    check("rn_conv dw", dw.cpu().numpy(), wt.grad.permute(2, 3, 1, 0).numpy())
    check("rn_conv dx", dx.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("B,H,aff,with_add,sums", [(2, 37, 1, 0, 1), (3, 8, 1, 0, 0), (1, 1, 0, 1, 0), (2, 50, 0, 0, 0), (2, 3, 0, 1, 0), (5, 13, 1, 0, 1)])
def test_xc_depthwise_backward(seld_lib, B, H, aff, with_add, sums, fused):
    """xception_block's depthwise 3x3 backward through seld_k_xc_dw_bwd, in ONE pass (fused = 1: input gradient + kernel-gradient slabs + the
    previous BatchNormalization's backward sums, dw3x3_w16_bwd_fused; round 5) and as the separate passes (fused = 0), against autograd in fp64:
    the unit is  a = relu(z scale + shift)  (or relu(x)),  y = DepthwiseConv2D(3, 'same', use_bias=False)(a),  L = sum(y dy):
    dx = dL/d(pre-ReLU input) (+ add), dk = dL/dk, sums = [sum dx | sum dx xhat].  Ragged sizes: one image row, rows that are not a multiple of the
    four a workgroup takes, several images (the halo must not cross an image boundary)."""
    rng = np.random.default_rng(B * 100 + H)
    shp = (B, H, 16, 64)
    xin = rng.standard_normal(shp).astype(np.float32)
    dy = rng.standard_normal(shp).astype(np.float32)
    k = (rng.standard_normal((3, 3, 64)) * 0.3).astype(np.float32)
    addv = rng.standard_normal(shp).astype(np.float32) if with_add else None
    sc = (rng.standard_normal(64) * 0.5 + 1.0).astype(np.float32)
    sh = (rng.standard_normal(64) * 0.3).astype(np.float32)
    mean = (rng.standard_normal(64) * 0.2).astype(np.float32)
    invstd = (rng.random(64) + 0.5).astype(np.float32)
    # fp64 reference by autograd
    z = torch.as_tensor(xin, dtype=torch.float64)
    pre = (z * torch.as_tensor(sc, dtype=torch.float64) + torch.as_tensor(sh, dtype=torch.float64)) if aff else z.clone()
    pre.requires_grad_(True)
    kt = torch.as_tensor(k, dtype=torch.float64).permute(2, 0, 1).reshape(64, 1, 3, 3).clone().requires_grad_(True)
    y = F.conv2d(torch.relu(pre).permute(0, 3, 1, 2), kt, padding=1, groups=64).permute(0, 2, 3, 1)
    (y * torch.as_tensor(dy, dtype=torch.float64)).sum().backward()
    dx_ref = pre.grad.numpy() + (addv.astype(np.float64) if with_add else 0.0)
    dk_ref = kt.grad.reshape(64, 3, 3).permute(1, 2, 0).numpy()
    xhat = (xin.astype(np.float64) - mean) * invstd
    sums_ref = np.concatenate([dx_ref.sum((0, 1, 2)), (dx_ref * xhat).sum((0, 1, 2))])
    d = lambda a: dev(a) if a is not None else None
    t = {n: d(v) for n, v in dict(dy=dy, k=k, xin=xin, add=addv, aff=np.concatenate([sc, sh]) if aff else None, mean=mean if sums else None,
                                  invstd=invstd if sums else None).items()}
    dx, dk, sm = (torch.full(sz, float("nan"), device="cuda") for sz in (shp, (3, 3, 64), (128,)))
    p = lambda a: ptr(a) if a is not None else None
    rc = seld_lib.seld_k_xc_dw_bwd(p(t["dy"]), p(t["k"]), p(t["xin"]), p(t["add"]), p(t["aff"]), p(t["mean"]), p(t["invstd"]), ptr(dx), ptr(dk),
                                   ptr(sm) if sums else None, B, H, fused)
    assert rc == 0
    check("xc dw bwd dx", dx.cpu().numpy(), dx_ref)
    check("xc dw bwd dk", dk.cpu().numpy(), dk_ref)
    if sums:
        check("xc dw bwd BatchNorm sums", sm.cpu().numpy(), sums_ref)


def test_xc_depthwise_backward_refuses_sums_without_the_folded_batchnorm(seld_lib):
    """the sums belong to the BatchNormalization folded into the loads: asked for without `aff` (or together with a residual gradient) -> INVALID"""
    z = torch.zeros(1, 4, 16, 64, device="cuda")
    k = torch.zeros(3, 3, 64, device="cuda")
    v = torch.zeros(128, device="cuda")
    for aff, add in ((None, None), (v, z)):
        rc = seld_lib.seld_k_xc_dw_bwd(ptr(z), ptr(k), ptr(z), ptr(add) if add is not None else None, ptr(aff) if aff is not None else None, ptr(v), ptr(v),
                                       ptr(z.clone()), ptr(k.clone()), ptr(v.clone()), 1, 4, 1)
        assert rc != 0


@pytest.mark.parametrize("npix,C,relu,with_res,with_mask", [(960, 32, 1, 0, 1), (480, 256, 1, 0, 1), (480, 1024, 1, 1, 1), (333, 96, 0, 0, 0),
                                                            (7680, 64, 1, 1, 1)])
def test_rn_bn_fwd_bwd(seld_lib, npix, C, relu, with_res, with_mask):
    """Training-mode BatchNormalization (+ residual, ReLU) of resnet50_block and its backward with the gradient gated by the
    ReLU's output, against float64 autograd."""
    rng = np.random.default_rng(32)
    z = (rng.standard_normal((npix, C)) * rng.uniform(0.5, 2.0, C) + rng.standard_normal(C)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((npix, C)).astype(np.float32) if with_res else None
    dy = rng.standard_normal((npix, C)).astype(np.float32)
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    gt, bt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mean, var = zt.mean(0), zt.var(0, unbiased=False)
    o = (zt - mean) * torch.rsqrt(var + 1e-3) * gt + bt
    if with_res:
        o = o + torch.tensor(res, dtype=torch.float64)
    if relu:
        o = torch.relu(o)
    out = torch.full((npix, C), float("nan"), device="cuda")
    zd, gd, bd, dyd = dev(z), dev(gamma), dev(beta), dev(dy)
    rd = dev(res) if with_res else None
    assert seld_lib.seld_k_rn_bn(ptr(zd), ptr(gd), ptr(bd), ptr(rd) if with_res else None, ptr(out), None, None, npix, C, relu) == 0
    check("rn_bn out", out.cpu().numpy(), o.detach().numpy())
    # backward: the product gates dy by (mask > 0) with mask = the layer's output; autograd's ReLU does the same
    (o * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    dz = torch.full((npix, C), float("nan"), device="cuda")
    dg, db = torch.full((C,), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda")
    use_mask = bool(relu and with_mask)
    assert seld_lib.seld_k_rn_bn_bwd(ptr(zd), ptr(dyd), ptr(out) if use_mask else None, ptr(gd), ptr(dz), ptr(dg), ptr(db), npix, C) == 0
    check("rn_bn dz", dz.cpu().numpy(), zt.grad.numpy())
    check("rn_bn dgamma", dg.cpu().numpy(), gt.grad.numpy())
    check("rn_bn dbeta", db.cpu().numpy(), bt.grad.numpy())


def _bf16(a):
    """round-to-nearest-even bf16 of an fp32 array, back as float64 (what the single-product kernels feed the matrix cores)"""
    return torch.as_tensor(np.asarray(a, np.float32)).bfloat16().double().numpy()


@pytest.mark.parametrize("B,H,W", [(2, 20, 16), (3, 10, 4), (1, 37, 16), (2, 45, 8)])
def test_bf16_single_product_conv64(seld_lib, B, H, W):
    """bf16 single-product mode (BASELINE configs[1] "bf16": seld_k_set_option("bf16_single", 1) / SELD_DTYPE_BF16): the 64 -> 64
    convolution kernels take ONE v_mfma_f32_32x32x16_bf16 product per fp32 product with both operands rounded to nearest-even bf16 and
    fp32 accumulation.  Exact statement of that arithmetic = the fp64 convolution of the bf16-ROUNDED operands: the kernel must agree
    with it to fp32 accumulation error (2e-6), forward, input gradient and kernel gradient; against the unrounded fp64 result the
    mode's own error (operand rounding, 2^-9 relative each) is what is printed."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, H, W, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, 64)) / 24).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    dz = rng.standard_normal((B, H, W, 64)).astype(np.float32)
    assert seld_lib.seld_k_set_option(b"bf16_single", 1) == 0
    try:
        z = torch.full((B, H, W, 64), float("nan"), device="cuda")
        st = torch.zeros(128, device="cuda")
        assert seld_lib.seld_k_conv3x3_fwd(ptr(dev(x)), ptr(dev(w)), ptr(dev(b)), ptr(z), ptr(st), B, H, W, 64, 64) == 0
        ref_r = _conv_ref(_bf16(x), _bf16(w), b)
        check(f"bf16-single conv_fwd {B,H,W} vs fp64 of rounded operands", z.cpu().numpy(), ref_r, tol=2e-6)
        print(f"[bf16] conv_fwd {B,H,W}: error of the mode against the unrounded fp64 result {rel_err(z.cpu().numpy(), _conv_ref(x, w, b)):.2e}")
        check("bf16-single conv_fwd sum(z)", st.cpu().numpy()[:64], ref_r.sum(axis=(0, 1, 2)), tol=1e-4 * np.sqrt(ref_r.size / 64))
        # input gradient = the same kernel on dz with flipped weights
        xg = torch.zeros((B, H, W, 64), dtype=torch.float64, requires_grad=True)
        wt = torch.as_tensor(_bf16(w)).permute(3, 2, 0, 1)
        y = F.conv2d(xg.permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1)
        (g,) = torch.autograd.grad(y, xg, torch.as_tensor(_bf16(dz)))
        dx = torch.full((B, H, W, 64), float("nan"), device="cuda")
        assert seld_lib.seld_k_conv3x3_dgrad(ptr(dev(dz)), ptr(dev(w)), ptr(dx), B, H, W, 64, 64) == 0
        check(f"bf16-single conv_dgrad {B,H,W} vs fp64 of rounded operands", dx.cpu().numpy(), g.numpy(), tol=2e-6)
    finally:
        seld_lib.seld_k_set_option(b"bf16_single", 0)
