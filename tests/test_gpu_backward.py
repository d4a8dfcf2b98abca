"""the backward kernels of the fusion network, operator by operator, against torch autograd on the CPU
(the arithmetic the reference's loss.backward() runs).  Every case drives the C ABI through autograd.Tape exactly as the
training step does.  Tolerance: max |hip - torch| <= 1e-3 * max(|torch|) per gradient tensor unless stated (measured
values are 1e-6 .. 1e-5)."""
import importlib
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def mod(name):
    return importlib.import_module("image-super-resolution_amd." + name)


def rel(got, want):
    return (got.cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-20)


def to_map(x):
    return mod("engine").nchw_to_map(x, DEV)


def to_nchw(m):
    return mod("engine").map_to_nchw(m)


def param(t):
    A = mod("autograd")
    v = t.detach().clone().to(DEV)
    return A.Param("p", v, torch.zeros_like(v))


def gen(seed=0):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------------------------------------- dense conv / linear
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
@pytest.mark.parametrize("B,H,W,Cin,N,k,act,bias", [
    (2, 17, 23, 12, 64, 3, "gelu", True),      # hierarchical stage conv, odd sizes
    (1, 64, 64, 128, 128, 3, "gelu", True),    # refine stack
    (2, 32, 32, 3, 32, 3, "relu", True),       # difficulty_net.0 (Cin 3 -> pad 4)
    (2, 32, 32, 32, 1, 3, "sigmoid", True),    # N = 1 (gates)
    (1, 40, 24, 76, 64, 3, "none", False),     # ResBlock conv without bias, Cin 76
    (2, 16, 16, 180, 128, 1, "none", True),    # align layer 1x1
    (1, 48, 48, 16, 4, 1, "none", True),       # freq_weight_conv.2
    (1, 9, 11, 96, 32, 3, "gelu", True),       # edge fusion.0 on a tiny map (M < 1536: the small-tile kernels)
])
def test_conv_backward(mode, B, H, W, Cin, N, k, act, bias):
    A, ops = mod("autograd"), mod("ops")
    ops.set_gemm_mode(mode)
    try:
        g = gen(B * H + N)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        b = torch.randn(N, generator=g) if bias else None
        dy = torch.randn(B, N, H, W, generator=g)
        acts = {"gelu": (ops.ACT_GELU, F.gelu), "relu": (ops.ACT_RELU, F.relu), "sigmoid": (ops.ACT_SIGMOID, torch.sigmoid),
                "none": (ops.ACT_NONE, lambda v: v)}
        xt, wt = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        bt = None if b is None else b.clone().requires_grad_(True)
        yt = acts[act][1](F.conv2d(xt, wt, bt, padding=k // 2))
        yt.backward(dy)
        t = A.Tape(DEV)
        pw, pb = param(w), (None if b is None else param(b))
        cp = A.ConvP(pw, pb, DEV)
        cp.repack()
        xv = A.Var(to_map(x))
        y = t.conv(xv, cp, act=acts[act][0])
        assert rel(to_nchw(y.v), yt.detach()) < 1e-3
        y.g = to_map(dy)
        t.backward()
        assert rel(to_nchw(xv.g), xt.grad) < 1e-3, "dgrad"
        assert rel(pw.g, wt.grad) < 1e-3, "wgrad"
        if b is not None:
            assert rel(pb.g, bt.grad) < 1e-3, "bias grad"
        # gradients accumulate: a second backward pass of the same step doubles the parameter gradients
        y2 = t.conv(A.Var(to_map(x)), cp, act=acts[act][0])
        y2.g = to_map(dy)
        t.backward()
        assert rel(pw.g, 2 * wt.grad) < 1e-3
    finally:
        ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))


def test_linear_backward_token_matrix():
    A, ops = mod("autograd"), mod("ops")
    g = gen(3)
    M, K, N = 4 * 16 * 16 * 9, 64, 192
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 8, torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xt, wt, bt = (v.clone().requires_grad_(True) for v in (x, w, b))
    F.linear(xt, wt, bt).backward(dy)
    t = A.Tape(DEV)
    pw, pb = param(w), param(b)
    cp = A.ConvP(pw, pb, DEV)
    cp.repack()
    xv = A.Var(x.to(DEV))
    y = t.linear(xv, cp)
    y.g = dy.to(DEV)
    t.backward()
    assert rel(xv.g, xt.grad) < 1e-3 and rel(pw.g, wt.grad) < 1e-3 and rel(pb.g, bt.grad) < 1e-3


def test_wgrad_large_pixel_count_is_split_deterministically():
    """2 x 256 x 256 pixels, 128 -> 128 3x3 (the refine stack at config 5's patch size): many pixel splits, exact f32 products"""
    hip, ops = mod("hip"), mod("ops")
    g = gen(5)
    x, dy = torch.randn(2, 128, 64, 256, generator=g), torch.randn(2, 128, 64, 256, generator=g)
    want = torch.nn.grad.conv2d_weight(x, (128, 128, 3, 3), dy, padding=1)
    xm, dm = to_map(x), to_map(dy)
    outs = []
    for _ in range(2):
        dw = torch.zeros(128, 128, 3, 3, device=DEV)
        part = torch.empty(1 << 24, device=DEV)
        db = torch.zeros(128, device=DEV)
        hip.call("ffsr_conv_wgrad_f32", xm.data_ptr(), 128, dm.data_ptr(), 128, dw.data_ptr(), db.data_ptr(), part.data_ptr(),
                 part.numel(), 2, 64, 256, 128, 128, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream)
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel(outs[0][0], want) < 1e-4 and rel(outs[0][1], dy.sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,N", [(2, 64, 256, 128, 128), (1, 37, 53, 128, 128), (1, 20, 64, 256, 128), (2, 5, 300, 128, 256),
                                         (3, 9, 7, 132, 200), (1, 260, 256, 76, 64), (1, 256, 257, 96, 32), (1, 256, 256, 132, 200)])
def test_wgrad_split_bf16_wide_layers(B, H, W, Cin, N):
    """ffsr_conv_wgrad_bf16x3: 3-wide layers whose channel counts are multiples of the 128 x 128 tile take the transposing-read
    bf16 kernel (three split products), everything about the contract unchanged: image borders (every row / column position of
    the taps), images narrower / wider than a 64-pixel chunk, ragged last chunk and split, several tiles, the bias gradient,
    run-to-run determinism; channel counts off the tile at HR pixel counts (76 -> 64, 96 -> 32: hierarchical_fusion.py, edge_enhancement.py)
    take the same kernel with the padding fragments skipped; 132 -> 200 on 189 pixels has no such kernel: exact fp32 path.
    Truth = fp64 autograd; the error of three-term split-bf16 products is ~1e-5 relative."""
    hip = mod("hip")
    g = gen(B * 1000 + W)
    x, dy = torch.randn(B, Cin, H, W, generator=g), torch.randn(B, N, H, W, generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (N, Cin, 3, 3), dy.double(), padding=1)
    xm, dm = to_map(x), to_map(dy)
    outs = []
    for _ in range(2):
        dw = torch.zeros(N, Cin, 3, 3, device=DEV)
        part = torch.empty(1 << 24, device=DEV)
        db = torch.zeros(N, device=DEV)
        hip.call("ffsr_conv_wgrad_bf16x3", xm.data_ptr(), xm.stride(2), dm.data_ptr(), dm.stride(2), dw.data_ptr(), db.data_ptr(),
                 part.data_ptr(), part.numel(), B, H, W, Cin, N, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream)
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel(outs[0][0].double(), want) < 3e-5 and rel(outs[0][1].double(), dy.double().sum((0, 2, 3))) < 1e-5

@pytest.mark.parametrize("B,H,W,Cin,N", [(2, 64, 256, 128, 128), (1, 37, 53, 128, 256), (1, 260, 256, 96, 128)])
def test_wgrad_from_planes_matches_fp32_input(B, H, W, Cin, N):
    """ffsr_conv_wgrad_bf16x3_planes: X given as the bf16 hi / lo planes the forward GEMM consumed -- the same staged bits as
    ffsr_conv_wgrad_bf16x3 makes of the fp32 map, so the results are bit-identical; shapes without a bf16 kernel are refused."""
    hip, ops = mod("hip"), mod("ops")
    g = gen(B + Cin + N)
    x, dy = torch.randn(B, Cin, H, W, generator=g), torch.randn(B, N, H, W, generator=g)
    xm, dm = to_map(x), to_map(dy)
    xp = ops.split_planes(xm)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    dp = ops.split_planes(dm) if N % 32 == 0 else None
    for planes in (0, 1, 2):
        if planes == 2 and dp is None:
            continue
        dw, db = torch.zeros(N, Cin, 3, 3, device=DEV), torch.zeros(N, device=DEV)
        part = torch.empty(1 << 24, device=DEV)
        if planes == 2:        # dY as planes too (what ffsr_act_bwd_planes_f32 writes)
            hip.call("ffsr_conv_wgrad_bf16x3_planes", xp.hi.data_ptr(), xp.lo.data_ptr(), xp.Cp, None, 0, dp.hi.data_ptr(),
                     dp.lo.data_ptr(), dp.Cp, dw.data_ptr(), db.data_ptr(), part.data_ptr(), part.numel(), B, H, W, Cin, N, 3, 3, 1, 1, st)
        elif planes == 1:
            hip.call("ffsr_conv_wgrad_bf16x3_planes", xp.hi.data_ptr(), xp.lo.data_ptr(), xp.Cp, dm.data_ptr(), dm.stride(2), None, None, 0,
                     dw.data_ptr(), db.data_ptr(), part.data_ptr(), part.numel(), B, H, W, Cin, N, 3, 3, 1, 1, st)
        else:
            hip.call("ffsr_conv_wgrad_bf16x3", xm.data_ptr(), xm.stride(2), dm.data_ptr(), dm.stride(2), dw.data_ptr(),
                     db.data_ptr(), part.data_ptr(), part.numel(), B, H, W, Cin, N, 3, 3, 1, 1, st)
        outs.append((dw.cpu(), db.cpu()))
    for o in outs[1:]:         # weights: the same staged bits -> bit-identical; bias: hi + lo of dY instead of dY (2^-17 relative)
        assert torch.equal(outs[0][0], o[0]) and rel(o[1], outs[0][1]) < 1e-5
    want = torch.nn.grad.conv2d_weight(x.double(), (N, Cin, 3, 3), dy.double(), padding=1)
    assert rel(outs[1][0].double(), want) < 3e-5
    with pytest.raises(Exception):      # 1x1: no bf16 weight-gradient kernel
        hip.call("ffsr_conv_wgrad_bf16x3_planes", xp.hi.data_ptr(), xp.lo.data_ptr(), xp.Cp, dm.data_ptr(), dm.stride(2), None, None, 0,
                 dw.data_ptr(), db.data_ptr(), part.data_ptr(), part.numel(), B, H, W, Cin, N, 1, 1, 0, 0, st)

@pytest.mark.parametrize("act,from_out", [(1, False), (2, True), (3, True)])
def test_act_bwd_planes_is_split_of_act_bwd(act, from_out):
    """ffsr_act_bwd_planes_f32 writes split(dy * act'(ref)) -- bit-identical to splitting ffsr_act_bwd_f32's fp32 result."""
    hip, ops, A = mod("hip"), mod("ops"), mod("autograd")
    g = gen(act)
    M, C = 5000, 128
    dy, ref = torch.randn(M, C, generator=g).to(DEV), torch.randn(M, C, generator=g).to(DEV)
    t = A.Tape(DEV)
    want = ops.split_planes(ops.as_map(t.act_bwd(dy, ref, act, 0.2, from_output=from_out)))
    pl = ops.Planes(1, 1, M, C, DEV)
    hip.call("ffsr_act_bwd_planes_f32", dy.data_ptr(), C, ref.data_ptr(), C, pl.hi.data_ptr(), pl.lo.data_ptr(), pl.Cp, M, C, act, 0.2,
             int(from_out), 1.0, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(pl.buf, want.buf)

@pytest.mark.parametrize("B,H,W,Cin,N", [(2, 96, 100, 128, 3), (1, 130, 131, 16, 1), (3, 80, 70, 32, 4), (1, 128, 129, 8, 2),
                                         (2, 96, 100, 3, 128), (1, 150, 113, 4, 32), (2, 90, 95, 1, 8), (1, 200, 90, 64, 3)])
def test_wgrad_thin_layers(B, H, W, Cin, N):
    """3x3 layers with N <= 4 outputs (refine.10, the edge / gate heads: enhanced_fusion_v2.py:576, edge_enhancement.py:88,177) or
    Cin <= 4 inputs (refine.0, enhanced_fusion_v2.py:569) at HR sizes: the vector-ALU outer-product kernels behind both weight-
    gradient entry points, exact fp32, against fp64 autograd; ragged rows / runs, every border, bias gradient, determinism."""
    hip = mod("hip")
    g = gen(B * 100 + Cin + N)
    x, dy = torch.randn(B, Cin, H, W, generator=g), torch.randn(B, N, H, W, generator=g)
    want = torch.nn.grad.conv2d_weight(x.double(), (N, Cin, 3, 3), dy.double(), padding=1)
    xm, dm = to_map(x), to_map(dy)
    for entry in ("ffsr_conv_wgrad_f32", "ffsr_conv_wgrad_bf16x3"):
        outs = []
        for _ in range(2):
            dw = torch.zeros(N, Cin, 3, 3, device=DEV)
            part = torch.empty(1 << 23, device=DEV)
            db = torch.zeros(N, device=DEV)
            hip.call(entry, xm.data_ptr(), xm.stride(2), dm.data_ptr(), dm.stride(2), dw.data_ptr(), db.data_ptr(),
                     part.data_ptr(), part.numel(), B, H, W, Cin, N, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream)
            outs.append((dw.cpu(), db.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert rel(outs[0][0].double(), want) < 2e-6 and rel(outs[0][1].double(), dy.double().sum((0, 2, 3))) < 2e-6

# ---------------------------------------------------------------------------------------------- depthwise / norms
@pytest.mark.parametrize("C,kh,kw,ph,pw_", [(64, 5, 5, 2, 2), (128, 1, 21, 0, 10), (64, 21, 1, 10, 0)])
def test_dwconv_backward(C, kh, kw, ph, pw_):
    A = mod("autograd")
    g = gen(C + kh)
    B, H, W = 2, 19, 27
    x, w, dy = torch.randn(B, C, H, W, generator=g), torch.randn(C, 1, kh, kw, generator=g), torch.randn(B, C, H, W, generator=g)
    xt, wt = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xt, wt, padding=(ph, pw_), groups=C).backward(dy)
    t = A.Tape(DEV)
    p = param(w)
    dw = A.DwP(p, DEV, (ph, pw_))
    xv = A.Var(to_map(x))
    y = t.dwconv(xv, dw)
    assert rel(to_nchw(y.v), F.conv2d(x, w, padding=(ph, pw_), groups=C)) < 1e-5
    y.g = to_map(dy)
    t.backward()
    assert rel(to_nchw(xv.g), xt.grad) < 1e-4 and rel(p.g, wt.grad) < 1e-4


@pytest.mark.parametrize("mean_over_std", [0.25, 50.0])
def test_batchnorm_train_forward_backward_and_running_stats(mean_over_std):
    """mean_over_std 50: un-normalised residual streams feed lka_global.norm1 -- the variance and dgamma are sums AROUND the
    batch mean (two-pass), so |mean| >> std does not cancel (checked against F.batch_norm evaluated in float64)"""
    A = mod("autograd")
    g = gen(9)
    B, C, H, W = 3, 64, 16, 24
    x = torch.randn(B, C, H, W, generator=g) * 1.7 + 1.7 * mean_over_std * (torch.rand(1, C, 1, 1, generator=g) + 0.5)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dy = torch.randn(B, C, H, W, generator=g)
    xt, gt, bt = (v.double().clone().requires_grad_(True) for v in (x, gamma, beta))
    rmt, rvt = rm.double().clone(), rv.double().clone()
    yt = F.batch_norm(xt, rmt, rvt, gt, bt, True, 0.1, 1e-5)
    yt.backward(dy.double())
    t = A.Tape(DEV)
    pg, pb = param(gamma), param(beta)
    bn = A.BnP(pg, pb, rm.to(DEV), rv.to(DEV))
    xv = A.Var(to_map(x))
    y = t.bn(xv, bn)
    # (at mean = 50 std the fp32 INPUT itself carries 50 x 6e-8 of relative noise in x - mean: 1e-5 stays the bar)
    assert rel(to_nchw(y.v), yt.detach().float()) < 1e-5
    assert rel(bn.run_mean, rmt.float()) < 1e-6 and rel(bn.run_var, rvt.float()) < 2e-6 and bn.calls == 1
    y.g = to_map(dy)
    t.backward()
    assert rel(to_nchw(xv.g), xt.grad.float()) < 1e-4 and rel(pg.g, gt.grad.float()) < 1e-4 and rel(pb.g, bt.grad.float()) < 1e-4


@pytest.mark.parametrize("M,C", [(2 * 16 * 16 * 9, 64), (1000, 128)])
def test_layernorm_backward(M, C):
    A = mod("autograd")
    g = gen(M)
    x, gamma, beta, dy = torch.randn(M, C, generator=g), torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.randn(M, C, generator=g)
    xt, gt, bt = (v.clone().requires_grad_(True) for v in (x, gamma, beta))
    F.layer_norm(xt, (C,), gt, bt).backward(dy)
    t = A.Tape(DEV)
    pg, pb = param(gamma), param(beta)
    xv = A.Var(x.to(DEV))
    y = t.layernorm(xv, pg, pb)
    y.g = dy.to(DEV)
    t.backward()
    assert rel(xv.g, xt.grad) < 1e-4 and rel(pg.g, gt.grad) < 1e-4 and rel(pb.g, bt.grad) < 1e-4


# ---------------------------------------------------------------------------------------------- resamplers
@pytest.mark.parametrize("Hi,Wi,Ho,Wo,C", [(16, 16, 64, 64, 32), (64, 64, 16, 16, 12), (64, 64, 32, 32, 12), (35, 51, 140, 204, 3),
                                           (64, 64, 32, 17, 1), (17, 13, 40, 29, 4), (8, 8, 1, 1, 3), (33, 47, 132, 188, 128), (20, 30, 50, 45, 8)])
def test_bilinear_adjoint(Hi, Wi, Ho, Wo, C):
    A = mod("autograd")
    g = gen(Hi + Wo)
    x, dy = torch.randn(2, C, Hi, Wi, generator=g), torch.randn(2, C, Ho, Wo, generator=g)
    xt = x.clone().requires_grad_(True)
    F.interpolate(xt, size=(Ho, Wo), mode="bilinear", align_corners=False).backward(dy)
    t = A.Tape(DEV)
    xv = A.Var(to_map(x))
    y = t.bilinear(xv, Ho, Wo)
    y.g = to_map(dy)
    t.backward()
    assert rel(to_nchw(xv.g), xt.grad) < 1e-5


def test_avgpool2_adjoint_odd_size():
    A = mod("autograd")
    g = gen(1)
    x, dy = torch.randn(2, 4, 9, 13, generator=g), torch.randn(2, 4, 4, 6, generator=g)
    xt = x.clone().requires_grad_(True)
    F.avg_pool2d(xt, 2, 2).backward(dy)
    t = A.Tape(DEV)
    xv = A.Var(to_map(x))
    y = t.avgpool2(xv)
    y.g = to_map(dy)
    t.backward()
    assert rel(to_nchw(xv.g), xt.grad) < 1e-6


# ---------------------------------------------------------------------------------------------- attention / tails
@pytest.mark.parametrize("T,E,heads", [(9, 64, 4), (4, 128, 8)])
def test_pixel_mha_backward(T, E, heads):
    A = mod("autograd")
    g = gen(T)
    S = 700
    qkv, dy = torch.randn(S * T, 3 * E, generator=g), torch.randn(S * T, E, generator=g)
    qt = qkv.clone().requires_grad_(True)
    q, k, v = (qt.reshape(S, T, 3, heads, 16)[:, :, i].transpose(1, 2) for i in range(3))
    o = ((q / 4.0) @ k.transpose(-2, -1)).softmax(-1) @ v
    o.transpose(1, 2).reshape(S * T, E).backward(dy)
    t = A.Tape(DEV)
    qv = A.Var(qkv.to(DEV))
    y = t.pixel_mha(qv, S, T, E, heads)
    y.g = dy.to(DEV)
    t.backward()
    assert rel(qv.g, qt.grad) < 1e-4


@pytest.mark.parametrize("T,E,heads", [(9, 64, 4), (4, 128, 8)])
def test_pixel_mha_attention_dropout(T, E, heads):
    """nn.MultiheadAttention(dropout=0.1) in train mode (large_kernel_attention.py:196,298): probabilities are dropped with a
    counter-based mask.  Checked: the mask is {0, 1/(1-p)}-valued with a keep rate within 3 sigma of 1-p, is a function of the
    seed only (same seed -> same output, other seed -> other mask, independent of the data), and the backward kernel
    differentiates exactly the masked attention of the forward pass (torch autograd on the recovered, frozen mask).
    Bit-parity with torch's Philox stream is unpinned by construction."""
    A, ops = mod("autograd"), mod("ops")
    p, S, g = 0.1, 900, gen(T + 100)
    qkv = torch.randn(S * T, 3 * E, generator=g)
    # recover the mask: V = one-hot of the key index in the first T dims of every head -> out[t, h, j] = P'[t, j]
    probe = qkv.clone().reshape(S, T, 3, heads, 16)
    probe[:, :, 2] = 0.0
    for j in range(T):
        probe[:, j, 2, :, j] = 1.0
    probe = probe.reshape(S * T, 3 * E)
    q, k = (probe.reshape(S, T, 3, heads, 16)[:, :, i].transpose(1, 2) for i in range(2))
    P = ((q / 4.0) @ k.transpose(-2, -1)).softmax(-1)                                      # [S, heads, T, T]
    out = ops.pixel_mha(probe.to(DEV), S, T, E, heads, p_drop=p, seed=0).cpu().reshape(S, T, heads, 16)[..., :T].permute(0, 2, 1, 3)
    mask = out / P                                                                          # 0 or 1 / (1 - p)
    kept = mask > 0.5
    assert torch.allclose(mask[kept], torch.full_like(mask[kept], 1.0 / (1.0 - p)), rtol=1e-4)
    assert float(mask[~kept].abs().max()) == 0.0
    n = mask.numel()
    assert abs(kept.float().mean().item() - (1 - p)) < 3.0 * (p * (1 - p) / n) ** 0.5
    # a function of the seed, not of the data
    again = ops.pixel_mha(probe.to(DEV), S, T, E, heads, p_drop=p, seed=0).cpu()
    assert torch.equal(again.reshape(S, T, heads, 16)[..., :T].permute(0, 2, 1, 3), out)
    other = ops.pixel_mha(probe.to(DEV), S, T, E, heads, p_drop=p, seed=1).cpu().reshape(S, T, heads, 16)[..., :T].permute(0, 2, 1, 3)
    assert 0.1 < ((other > 0) != kept).float().mean().item() < 0.3                         # ~ 2 p (1 - p) = 0.18 of the entries differ
    # forward on real data = masked attention; backward = its gradient
    dy = torch.randn(S * T, E, generator=g)
    qt = qkv.clone().requires_grad_(True)
    q, k, v = (qt.reshape(S, T, 3, heads, 16)[:, :, i].transpose(1, 2) for i in range(3))
    o = (((q / 4.0) @ k.transpose(-2, -1)).softmax(-1) * mask) @ v
    want = o.transpose(1, 2).reshape(S * T, E)
    want.backward(dy)
    t = A.Tape(DEV)                      # seed 0, step 0, first draw -> the probe's seed 0
    qv = A.Var(qkv.to(DEV))
    y = t.pixel_mha(qv, S, T, E, heads, p_drop=p)
    assert rel(y.v, want.detach()) < 1e-5
    y.g = dy.to(DEV)
    t.backward()
    assert rel(qv.g, qt.grad) < 1e-4
    # p = 0 stays the eval kernel bit for bit
    assert torch.equal(ops.pixel_mha(qkv.to(DEV), S, T, E, heads), ops.pixel_mha(qkv.to(DEV), S, T, E, heads, p_drop=0.0, seed=5))


def test_softmax_expert_sum_selector_backward():
    A = mod("autograd")
    g = gen(2)
    B, H, W = 2, 24, 20
    logits, enh = torch.randn(B, 4, H, W, generator=g), torch.rand(B, 12, H, W, generator=g)
    gates = torch.rand(B, 4, H, W, generator=g) * 0.2          # small sums: the clamp(min=0.3) of the selector is active on some pixels
    dy = torch.randn(B, 3, H, W, generator=g)
    lt, et, gt = (v.clone().requires_grad_(True) for v in (logits, enh, gates))
    wts = lt.softmax(1)
    freq = sum(et[:, 3 * e:3 * e + 3] * wts[:, e:e + 1] for e in range(4))
    dyn = sum(et[:, 3 * e:3 * e + 3] * gt[:, e:e + 1] for e in range(4)) / (gt.sum(1, keepdim=True) + 1e-8)
    (freq + 2 * dyn).backward(dy)
    t = A.Tape(DEV)
    lv, ev, gv = A.Var(to_map(logits)), A.Var(to_map(enh)), A.Var(to_map(gates))
    y = t.add(t.expert_sum(ev, t.softmax_c(lv), normalize=False), t.expert_sum(ev, gv, normalize=True), 1.0, 2.0)
    y.g = to_map(dy)
    t.backward()
    assert rel(to_nchw(lv.g), lt.grad) < 1e-4 and rel(to_nchw(ev.g), et.grad) < 1e-4 and rel(to_nchw(gv.g), gt.grad) < 1e-4
    # DynamicExpertSelector tail (enhanced_fusion_v2.py:462-465)
    raw, d = torch.randn(B, 4, H, W, generator=g) * 0.3 + 0.5, torch.rand(B, 1, H, W, generator=g)
    T = torch.tensor(10.0)
    dg = torch.randn(B, 4, H, W, generator=g)
    rt, dt, Tt = raw.clone().requires_grad_(True), d.clone().requires_grad_(True), T.clone().requires_grad_(True)
    s = torch.sigmoid(Tt * (rt - (0.7 - 0.5 * dt)))
    (s / (s.sum(1, keepdim=True) + 1e-8).clamp(min=0.3)).backward(dg)
    t = A.Tape(DEV)
    pT = param(T)
    rv, dv = A.Var(to_map(raw)), A.Var(to_map(d))
    y = t.selector_gates(rv, dv, pT)
    y.g = to_map(dg)
    t.backward()
    assert rel(to_nchw(rv.g), rt.grad) < 1e-4 and rel(to_nchw(dv.g), dt.grad) < 1e-4 and rel(pT.g, Tt.grad) < 1e-4


def test_elementwise_ops_backward():
    """mul with a per-pixel gate, learnable residual scale, GELU / clamp, views (join / split)"""
    A, ops = mod("autograd"), mod("ops")
    g = gen(4)
    B, C, H, W = 2, 32, 12, 10
    x, gate = torch.randn(B, C, H, W, generator=g), torch.rand(B, 1, H, W, generator=g)
    s = torch.tensor(0.37)
    dy = torch.randn(B, C, H, W, generator=g)
    xt, gt, st = x.clone().requires_grad_(True), gate.clone().requires_grad_(True), s.clone().requires_grad_(True)
    xg = xt * gt
    r = F.gelu(xg)
    out = (xg + st * r).clamp(0, 1) + 0.5 * xt[:, :C] * 1.0
    out.backward(dy)
    t = A.Tape(DEV)
    ps = param(s)
    xv, gv = A.Var(to_map(x)), A.Var(to_map(gate))
    xg_ = t.mul(xv, gv, row_broadcast=True)
    pre = t.add_scaled(xg_, t.act(xg_, ops.ACT_GELU), ps.v, ps.g)
    y = t.add(t.act(pre, A.ACT_CLAMP01), xv, 1.0, 0.5)
    y.g = to_map(dy)
    t.backward()
    assert rel(to_nchw(xv.g), xt.grad) < 1e-4 and rel(to_nchw(gv.g), gt.grad) < 1e-4 and rel(ps.g, st.grad) < 1e-4
    # join (concat by channel slices of one buffer) and split (token-interleaved rows)
    a, b = torch.randn(B, 8, H, W, generator=g), torch.randn(B, 4, H, W, generator=g)
    at, bt = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    cat = torch.cat([at * 2.0, bt * 3.0], 1)
    dcat = torch.randn(B, 12, H, W, generator=g)
    (cat * cat).backward(dcat)
    t = A.Tape(DEV)
    buf = torch.empty(B, H, W, 12, device=DEV)
    av, bv = A.Var(to_map(a)), A.Var(to_map(b))
    pa, pb = t.affine(av, 2.0, 0.0, out=buf[..., :8]), t.affine(bv, 3.0, 0.0, out=buf[..., 8:12])
    whole = t.join([pa, pb], buf, [A.ch(0, 8), A.ch(8, 12)])
    y = t.mul(whole, whole)
    y.g = to_map(dcat)
    t.backward()
    assert rel(to_nchw(av.g), at.grad) < 1e-5 and rel(to_nchw(bv.g), bt.grad) < 1e-5
    T_, E_ = 4, 16
    m = torch.randn(B * H * W * T_, E_, generator=g)
    mt = m.clone().requires_grad_(True)
    parts_t = mt.reshape(B, H, W, T_, E_)
    dparts = [torch.randn(B, E_, H, W, generator=g) for _ in range(2)]
    (parts_t[:, :, :, 0].permute(0, 3, 1, 2) * dparts[0]).sum().backward(retain_graph=True)
    (parts_t[:, :, :, 2].permute(0, 3, 1, 2) * dparts[1]).sum().backward()
    t = A.Tape(DEV)
    mv = A.Var(m.to(DEV))
    parts = t.split(mv, [A.tok(e, T_, B, H, W) for e in range(T_)])
    parts[0].g, parts[2].g = to_map(dparts[0]), to_map(dparts[1])
    t.backward()
    assert rel(mv.g, mt.grad) < 1e-6


def test_pack_conv_matches_load_time_packing():
    """the per-step repacking kernel against ops.pack_conv (torch ops at load time): forward operand bit-identical;
    the transposed operand reproduces conv2d's input gradient (covered by test_conv_backward)"""
    A, ops = mod("autograd"), mod("ops")
    g = gen(6)
    for (N, Cin, k, cin_pad) in ((64, 12, 3, None), (128, 3, 3, 4), (3, 128, 3, None), (192, 64, 1, None)):
        w = torch.randn(N, Cin, k, k, generator=g)
        ref = ops.pack_conv(w, None, DEV, cin_pad=cin_pad)
        cp = A.ConvP(param(w), None, DEV, cin_pad=cin_pad)
        cp.repack()
        rows = ref.whi.shape[0]
        assert torch.equal(cp.fwd.wgt, ref.wgt) and torch.equal(cp.fwd.whi[:rows], ref.whi) and torch.equal(cp.fwd.wlo[:rows], ref.wlo)
        assert cp.fwd.whi[rows:].float().abs().max().item() == 0 if cp.fwd.whi.shape[0] > rows else True
        if cp.fwd.phi is not None:      # Cin % 32 == 0: the same buffers are the planes kernels' weight operand
            assert cp.fwd.Cp32 == ref.Cp32 and torch.equal(cp.fwd.phi, ref.phi) and torch.equal(cp.fwd.plo, ref.plo)


# ---------------------------------------------------------------------------------------------- the whole network
def _train_inputs(case):
    E = mod("engine")
    lr, hr = E.nchw_to_map(case["lr"], DEV), E.nchw_to_map(case["hr"], DEV)
    imgs = {k: E.nchw_to_map(v.float(), DEV) for k, v in case["imgs"].items()}
    feats = {k: E.nchw_to_map(v.float(), DEV) for k, v in case["feats"].items()}
    return lr, hr, imgs, feats


def _grad_report(trainer, want_grads, tol, floor=0.0):
    got = trainer.opt.views(trainer.opt.grad)
    worst, bad = [], []
    for k, w in want_grads.items():
        e = (got[k].cpu() - w).abs().max().item()
        r = e / max(w.abs().max().item(), 1e-30)
        worst.append((r, k, w.abs().max().item()))
        if r > tol and e > floor:
            bad.append((k, r, e, w.abs().max().item()))
    worst.sort(reverse=True)
    print("worst per-tensor relative gradient errors:", [(f"{r:.1e}", k, f"|g|max {m:.1e}") for r, k, m in worst[:6]])
    return bad


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_fusion_train_step_against_reference_fixture(mode):
    """tests/golden/fusion_train.pt = the REFERENCE's model.train() forward + loss.backward() (dropout 0) on a seeded batch
    of two 32x32 tiles (oracle/make_golden.py golden_train): sr, loss, the gradient of every one of the 198 parameter
    tensors (1.43 M values) and the BatchNorm running statistics after the forward.  Per-tensor tolerance 1e-3 relative to
    max|reference gradient|; gradients whose largest entry is below 1e-9 (the FFT temperature: 1e-11, noise level of fp32
    autograd itself -- the CPU oracle differs from the reference by 6e-4 there) are checked to 1e-12 absolute instead."""
    from conftest import load_golden
    T, ops = mod("train"), mod("ops")
    case, sd = load_golden("fusion_train.pt"), load_golden("fusion_full.pt")["sd"]
    ops.set_gemm_mode(mode)
    try:
        tr = T.FusionTrainer(sd, DEV, attn_dropout=0.0)
        lr, hr, imgs, feats = _train_inputs(case)
        tr.zero_grad()
        loss, sr = tr.forward_backward(lr, hr, imgs, feats)
        E = mod("engine")
        e_sr = (E.map_to_nchw(sr) - case["sr"]).abs().max().item()
        print(f"[{mode}] train-mode sr max abs err {e_sr:.2e}, loss {loss.item():.7f} vs {case['loss'].item():.7f}")
        assert e_sr < 1e-3 and abs(loss.item() - case["loss"].item()) < 1e-5
        for k, v in case["stats"].items():
            assert rel(tr.buffers[k], v) < 1e-4, k
        bad = _grad_report(tr, case["grads"], 1e-3, floor=1e-12)
        assert not bad, bad
        assert all(bn.calls == (9 if i < 3 else 4) for i, bn in enumerate(tr.net.bn_modules()))
    finally:
        ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_fusion_gradients_and_full_step_vs_oracle_autograd_64x64(mode):
    """B = 2 at 64x64 -- HIP gradients of all 1.43 M parameters against torch-CPU autograd through
    oracle/ffsr_oracle/fusion.py (train mode), then ONE full step (loss -> grads -> clip_grad_norm_ -> AdamW -> EMA,
    train.py:323-359) against oracle/ffsr_oracle/train.py's Trainer fed with the oracle's gradients.
    Modes: f32 = exact f32-MFMA convolutions (the arithmetic class of the reference's CPU path) -> 1e-3 per tensor;
    bf16x3 (default) = 1e-5-relative products: a ReLU / clamp whose argument is within that distance of 0 flips its mask,
    and one flipped unit of the selector's 3x3 convs moves a weight gradient by ~1 / (number of pixels) -> 3e-3 per tensor
    (measured: 1.9e-3 on dynamic_selector.gate_net.0.weight, everything else <= 1.5e-3)."""
    from ffsr_oracle import fusion as ofusion, train as otrain
    from make_golden import train_case
    T, W, FT, ops = mod("train"), mod("weights"), mod("fusion_train"), mod("ops")
    ops.set_gemm_mode(mode)
    try:
        _full_step_case(mode, ofusion, otrain, train_case, T, W, FT)
    finally:
        ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))


def _full_step_case(mode, ofusion, otrain, train_case, T, W, FT):
    sd = {k: v for k, v in W.fusion_state_dict(seed=5).items() if v.is_floating_point() and v.numel() > 0}
    lr, imgs, feats, hr = train_case(77, 2, 64, 64)
    case = {"lr": lr, "hr": hr, "imgs": imgs, "feats": feats}
    names = [k for k in sd if FT.is_parameter(k)]

    def oracle_grads(dt):
        s_ = {k: (v.clone().to(dt).requires_grad_(True) if k in names else v.clone().to(dt)) for k, v in sd.items()}
        sr_ = ofusion.fusion_forward(s_, lr.to(dt), {k: v.to(dt) for k, v in imgs.items()}, {k: v.to(dt) for k, v in feats.items()},
                                     train=True)
        l_ = F.l1_loss(sr_.clamp(0, 1), hr.to(dt))
        l_.backward()
        return l_.item(), {k: s_[k].grad for k in names}

    # With these weights several gradients are sums with heavy cancellation (|g| ~ 1e-7): torch's OWN fp32 autograd is up to
    # 1.6e-3 (relative, per tensor) away from the float64 evaluation of the same graph.  So the truth is the float64 oracle,
    # and the HIP gradient may deviate from it by 1e-3 plus twice what fp32 CPU autograd (the reference's arithmetic) does.
    loss_o, want = oracle_grads(torch.float32)
    _, truth = oracle_grads(torch.float64)
    tr = T.FusionTrainer(sd, DEV, attn_dropout=0.0)
    loss = tr.step(*_train_inputs(case))
    assert abs(loss.item() - loss_o) < 1e-5
    got = tr.opt.views(tr.opt.grad)
    rows = []
    for k in names:
        scale = max(truth[k].abs().max().item(), 1e-30)
        e_hip = (got[k].cpu().double() - truth[k]).abs().max().item() / scale
        e_cpu = (want[k].double() - truth[k]).abs().max().item() / scale
        rows.append((e_hip, e_cpu, k, scale))
    rows.sort(reverse=True)
    print("worst per-tensor relative gradient error vs the float64 oracle (hip, torch-fp32-cpu):",
          [(f"{a:.1e}", f"{b:.1e}", k, f"|g|max {m:.1e}") for a, b, k, m in rows[:6]])
    tol = 1e-3 if mode == "f32" else 3e-3
    bad = [(k, a, b) for a, b, k, m in rows if a > tol + 2 * b and a * m > 1e-12]
    assert not bad, bad
    # (a) the optimiser pipeline itself: torch's clip_grad_norm_ + AdamW + EMA fed with the HIP gradients -> tight agreement
    hip_grads = {k: v.cpu().clone() for k, v in tr.opt.views(tr.opt.grad).items()}
    got_p, got_e = tr.opt.views(), tr.opt.views(tr.opt.ema)
    ref = otrain.Trainer({k: sd[k] for k in names})
    ref.step(hip_grads)
    for k, p in ref.params.items():
        assert (got_p[k].cpu() - p.data).abs().max().item() <= 2e-6 * max(1.0, p.data.abs().max().item()), k
        assert (got_e[k].cpu() - ref.shadow[k]).abs().max().item() <= 2e-6 * max(1.0, p.data.abs().max().item()), k
    # (b) the whole step against the oracle's own gradients.  Adam's first update is lr * g / (|g| + eps): an element whose
    # gradient is at rounding-noise level can move by up to 2 lr = 4e-4 differently, everything else agrees to ~1e-7
    ref = otrain.Trainer({k: sd[k] for k in names})
    ref.step(want)
    dev = torch.cat([(got_p[k].cpu() - p.data).abs().reshape(-1) for k, p in ref.params.items()])
    q = torch.quantile(dev[torch.randperm(dev.numel(), generator=gen(0))[:200000]], torch.tensor([0.5, 0.9, 0.99]))
    print(f"full step vs oracle: |dparam| median {q[0]:.1e}, p90 {q[1]:.1e}, p99 {q[2]:.1e}, max {dev.max():.1e}")
    assert dev.max().item() <= 4.1e-4 and q[1].item() <= 1e-5
    # the weights the next forward uses are the updated ones (packed operands follow the flat buffer)
    cp = tr.net.refine[2]
    assert torch.equal(cp.fwd.wgt.cpu()[:, :128], got_p["refine.4.weight"].cpu().permute(0, 2, 3, 1).reshape(128, -1)[:, :128])


def test_training_step_at_baseline_batch_32():
    """BASELINE config 5's geometry: ONE step on a batch of 32 patches of 64x64 (2.1 M HR pixels: the pixel-split paths of the
    weight-gradient kernels, the 2^21-row reductions).  The batch is 16 copies of two seeded patches, so the batch statistics of
    the BatchNorms, the mean L1 loss and every mean gradient equal those of the 2-patch batch -- which the CPU oracle
    (oracle/ffsr_oracle/fusion.py in train mode + torch autograd, float64) can evaluate: loss and the gradients of refine.4.weight,
    the cross-band attention's projections and dynamic_selector.gate_net.0.weight are checked against it.  Attention dropout 0
    (the oracle has no counter-based mask)."""
    from ffsr_oracle import fusion as ofusion
    from make_golden import train_case
    T, W, FT, E = mod("train"), mod("weights"), mod("fusion_train"), mod("engine")
    sd = {k: v for k, v in W.fusion_state_dict(seed=5).items() if v.is_floating_point() and v.numel() > 0}
    lr, imgs, feats, hr = train_case(78, 2, 64, 64)
    keys = ["refine.4.weight", "cross_band.band_attention.in_proj_weight", "cross_band.band_attention.out_proj.weight",
            "cross_band.band_proj.weight", "dynamic_selector.gate_net.0.weight"]
    s_ = {k: (v.clone().double().requires_grad_(True) if k in keys else v.clone().double()) for k, v in sd.items()}
    sr_ = ofusion.fusion_forward(s_, lr.double(), {k: v.double() for k, v in imgs.items()}, {k: v.double() for k, v in feats.items()},
                                 train=True)
    l_ = F.l1_loss(sr_.clamp(0, 1), hr.double())
    l_.backward()
    rep = lambda t: t.repeat(16, 1, 1, 1)
    tr = T.FusionTrainer(sd, DEV, attn_dropout=0.0)
    tr.zero_grad()
    loss, sr = tr.forward_backward(E.nchw_to_map(rep(lr), DEV), E.nchw_to_map(rep(hr), DEV), {k: E.nchw_to_map(rep(v), DEV) for k, v in imgs.items()},
                                   {k: E.nchw_to_map(rep(v), DEV) for k, v in feats.items()})
    assert tuple(sr.shape) == (32, 256, 256, 3)
    assert abs(loss.item() - l_.item()) < 1e-5
    got = tr.opt.views(tr.opt.grad)
    for k in keys:
        w = s_[k].grad
        r = (got[k].cpu().double() - w).abs().max().item() / max(w.abs().max().item(), 1e-30)
        print(f"B = 32 step: {k}: relative gradient error {r:.2e} (|g|max {w.abs().max().item():.1e})")
        assert r < 3e-3, (k, r)
    # every copy of a patch gives the same output rows
    srn = E.map_to_nchw(sr)
    assert (srn[0] - srn[2]).abs().max().item() < 1e-6 and (srn[1] - srn[31]).abs().max().item() < 1e-6


def test_training_odd_size_and_gradient_accumulation():
    """ragged patch size (20x28 LR: non-power-of-2 DFT, reflect-padded DCT blocks, odd resampler ratios in the hierarchical and
    Laplacian pyramids) in the exact mode against the float64 oracle; and accumulation_steps = 2 (train.py:332-345): two
    micro-batches accumulate loss / 2 gradients, the optimiser steps once, on the second call."""
    from ffsr_oracle import fusion as ofusion
    from make_golden import train_case
    T, W, FT, ops = mod("train"), mod("weights"), mod("fusion_train"), mod("ops")
    sd = {k: v for k, v in W.fusion_state_dict(seed=9).items() if v.is_floating_point() and v.numel() > 0}
    names = [k for k in sd if FT.is_parameter(k)]
    batches = [train_case(81, 1, 20, 28), train_case(82, 1, 20, 28)]
    dt = torch.float64
    s_ = {k: (v.clone().to(dt).requires_grad_(True) if k in names else v.clone().to(dt)) for k, v in sd.items()}
    for lr, imgs, feats, hr in batches:          # BatchNorm running statistics advance between the micro-batches, as on the device
        sr = ofusion.fusion_forward(s_, lr.to(dt), {k: v.to(dt) for k, v in imgs.items()}, {k: v.to(dt) for k, v in feats.items()},
                                    train=True)
        (F.l1_loss(sr.clamp(0, 1), hr.to(dt)) / 2).backward()
    truth = {k: s_[k].grad for k in names}
    ops.set_gemm_mode("f32")
    try:
        tr = T.FusionTrainer(sd, DEV, accumulation_steps=2, attn_dropout=0.0)
        p0 = tr.opt.param.clone()
        for i, (lr, imgs, feats, hr) in enumerate(batches):
            tr.step(*_train_inputs({"lr": lr, "hr": hr, "imgs": imgs, "feats": feats}))
            if i == 0:
                assert torch.equal(tr.opt.param, p0) and tr.opt.step_count == 0        # no optimiser step yet
        assert tr.opt.step_count == 1 and not torch.equal(tr.opt.param, p0)
        got = tr.opt.views(tr.opt.grad)
        bad = []
        for k in names:
            scale = max(truth[k].abs().max().item(), 1e-30)
            e = (got[k].cpu().double() - truth[k]).abs().max().item()
            if e > 3e-3 * scale and e > 1e-12:
                bad.append((k, e / scale, scale))
        assert not bad, bad
    finally:
        ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))


def test_training_reduces_the_loss_on_a_fixed_batch():
    """end-to-end sanity of the whole step: 25 AdamW steps (lr 1e-3) on one fixed batch must lower the clamped-L1 loss, the EMA
    shadow must trail the weights, and the trained weights must load back into the inference engine's fusion net"""
    from make_golden import train_case
    from conftest import load_golden
    T, F_ = mod("train"), mod("fusion")
    sd = load_golden("fusion_full.pt")["sd"]
    lr, imgs, feats, hr = train_case(91, 2, 32, 32)
    inputs = _train_inputs({"lr": lr, "hr": hr, "imgs": imgs, "feats": feats})
    tr = T.FusionTrainer(sd, DEV, lr=1e-3, attn_dropout=0.0)
    losses = [tr.step(*inputs).item() for _ in range(25)]
    print("loss: first 3", [f"{v:.5f}" for v in losses[:3]], "last 3", [f"{v:.5f}" for v in losses[-3:]])
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < 0.9 * losses[0], losses
    p, e = tr.opt.views(), tr.opt.views(tr.opt.ema)
    k = "refine.4.weight"
    assert (p[k].cpu() - sd[k]).abs().max() > (e[k].cpu() - sd[k]).abs().max() > 0          # EMA (0.999) lags behind
    # the trained state_dict (reference keys) drives the inference network
    net = F_.FusionNet({k_: v.cpu() for k_, v in tr.state_dict().items()}, DEV)
    lrm, _, im, ft = inputs
    out = net(lrm, im, ft)
    assert torch.isfinite(out).all() and 0.0 <= out.min().item() and out.max().item() <= 1.0
