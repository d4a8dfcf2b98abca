"""GPU: every C-ABI kernel against a plain PyTorch fp32 (CPU) computation of the same op.

Tolerances are fp32 round-off scaled by the contraction length (the f32 MFMA is an exact fmaf chain, the CPU
reference sums in a different order)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E(pkg):
    import importlib
    return importlib.import_module("image-super-resolution_amd.engine")


@pytest.fixture(scope="module")
def ops(pkg):
    import importlib
    return importlib.import_module("image-super-resolution_amd.ops")


DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def gemm_tol(ops):
    return 2e-5 if ops.GEMM_MODE == "f32" else 2e-4


def close(got, want, tol, what=""):
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= tol * max(1.0, ref), f"{what}: max err {err:.3e} (ref max {ref:.3e}, tol {tol})"


ACT = {0: lambda v: v, 1: F.gelu, 2: F.relu, 3: lambda v: F.leaky_relu(v, 0.2), 4: torch.sigmoid, 5: F.silu}


@pytest.mark.parametrize("B,H,W,Cin,N,k,stride,act,hint", [
    (1, 16, 16, 180, 180, 1, 1, 0, 0), (2, 9, 13, 64, 96, 3, 1, 1, 0), (1, 20, 24, 12, 64, 3, 1, 1, 1),
    (1, 16, 16, 3, 32, 3, 1, 2, 0), (1, 16, 18, 45, 180, 3, 1, 0, 2), (1, 12, 12, 64, 128, 2, 2, 0, 0),
    (1, 33, 35, 128, 3, 3, 1, 4, 0), (1, 8, 8, 308, 180, 1, 1, 3, 4), (1, 64, 64, 180, 540, 1, 1, 0, 1),
    (3, 1, 1, 180, 10, 1, 1, 2, 0),
    # thin outputs on large maps: the 32-column tile of the split-bf16 kernel
    (1, 64, 48, 64, 3, 3, 1, 0, 0), (1, 40, 52, 32, 16, 3, 1, 1, 0), (2, 64, 64, 12, 32, 3, 1, 2, 0), (1, 64, 64, 128, 1, 1, 1, 4, 0),
])
def test_conv2d(ops, E, B, H, W, Cin, N, k, stride, act, hint):
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(N, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k)), rnd(N, seed=3)
    pad = 0 if k == 2 else k // 2
    want = ACT[act](F.conv2d(x, w, b, stride=stride, padding=pad))
    cv = ops.pack_conv(w, b, DEV, stride=stride, pad=pad)
    xm = E.nchw_to_map(x, DEV)
    got = E.map_to_nchw(ops.conv2d(ops.widen(xm, cv.Cin), cv, act=act, slope=0.2, tile_hint=hint))
    close(got, want, gemm_tol(ops), "conv2d")


def test_conv2d_epilogue_residual_scale_shuffle_akscale(ops, E):
    B, H, W, C, N = 2, 10, 12, 64, 128
    x, w, b = rnd(B, C, H, W, seed=1), rnd(N, C, 1, 1, seed=2, scale=0.1), rnd(N, seed=3)
    res, cvec, rvec, ak = rnd(B, N, H, W, seed=4), rnd(N, seed=5), rnd(N, seed=6), rnd(B, C, seed=7)
    want = res * rvec[None, :, None, None] * 0.5 + F.conv2d(x * ak[:, :, None, None], w, b) * cvec[None, :, None, None] * 2.0
    cv = ops.pack_conv(w, b, DEV)
    got = ops.conv2d(E.nchw_to_map(x, DEV), cv, res=E.nchw_to_map(res, DEV), cvec=cvec.to(DEV), rvec=rvec.to(DEV),
                     cscale=2.0, rscale=0.5, akscale=ak.to(DEV))
    close(E.map_to_nchw(got), want, gemm_tol(ops), "epilogue")
    # fused PixelShuffle(2) with a residual at the shuffled position (NAFNet ups + skip)
    skip = rnd(B, N // 4, 2 * H, 2 * W, seed=8)
    want = F.pixel_shuffle(F.conv2d(x, w), 2) + skip
    got = ops.conv2d(E.nchw_to_map(x, DEV), ops.pack_conv(w, None, DEV), shuffle=2, res=E.nchw_to_map(skip, DEV))
    close(E.map_to_nchw(got), want, gemm_tol(ops), "shuffle")


@pytest.mark.parametrize("B,H,W,c", [(1, 40, 52, 64), (2, 33, 31, 128), (1, 64, 64, 32)])
def test_conv2d_fused_simple_gate(ops, E, B, H, W, c):
    """conv (c -> 2c) + SimpleGate (x1 * x2, nafnet_arch.py:21-24) + scaled residual in one kernel (store mode 3)."""
    if ops.GEMM_MODE != "bf16x3":
        pytest.skip("the gate store exists on the split-bf16 kernel only")
    x, w, b = rnd(B, c, H, W, seed=1), rnd(2 * c, c, 1, 1, seed=2, scale=1 / math.sqrt(c)), rnd(2 * c, seed=3)
    res, cvec = rnd(B, c, H, W, seed=4), rnd(c, seed=5)
    t = F.conv2d(x, w, b)
    want = res + t[:, :c] * t[:, c:] * cvec[None, :, None, None]
    cv = ops.pack_conv(w, b, DEV, gate_pairs=True)
    got = ops.conv2d(E.nchw_to_map(x, DEV), cv, gate=True, res=E.nchw_to_map(res, DEV), cvec=cvec.to(DEV))
    assert tuple(got.shape) == (B, H, W, c)
    close(E.map_to_nchw(got), want, 2e-4, "fused gate")
    plain = ops.pack_conv(w, b, DEV)
    t2 = ops.conv2d(E.nchw_to_map(x, DEV), plain, tile_hint=128)
    g2 = ops.mul_add(t2[..., :c], t2[..., c:])
    close(ops.conv2d(E.nchw_to_map(x, DEV), cv, gate=True), g2, 1e-6, "fused vs separate gate")


def test_conv2d_strided_views(ops, E):
    """inputs / outputs that are channel slices of wider buffers (dense-concat buffers of DRCT / hierarchical fusion)"""
    P, wide = 200, 308
    buf = rnd(P, wide, seed=1).to(DEV)
    w, b = rnd(32, 212, seed=2, scale=0.05), rnd(32, seed=3)
    want = F.leaky_relu(F.linear(buf[:, :212].cpu(), w, b), 0.2)
    ops.linear(buf[:, :212], ops.pack_conv(w, b, DEV), act=3, slope=0.2, out=buf[:, 212:244])
    close(buf[:, 212:244].cpu(), want, gemm_tol(ops), "slice out")


@pytest.mark.parametrize("C", [64, 180, 308, 1024, 3])
def test_layernorm(ops, C):
    x, g, b, r1, r2 = rnd(77, C, seed=1), rnd(C, seed=2), rnd(C, seed=3), rnd(77, C, seed=4), rnd(77, C, seed=5)
    got = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), eps=1e-6, res1=r1.to(DEV), res2=r2.to(DEV))
    close(got.cpu(), F.layer_norm(x, (C,), g, b, 1e-6) + r1 + r2, 1e-5, "layernorm")


def test_elementwise(ops, E):
    a, b = rnd(2, 24, 7, 9, seed=1), rnd(2, 24, 7, 9, seed=2)
    am, bm = E.nchw_to_map(a, DEV), E.nchw_to_map(b, DEV)
    cs, cb = rnd(24, seed=3), rnd(24, seed=4)
    close(E.map_to_nchw(ops.unary(am, act=1, pre=0.5, alpha=2.0, beta=0.1, cscale=cs.to(DEV), cbias=cb.to(DEV))),
          F.gelu(a * 0.5) * 2.0 * cs[None, :, None, None] + 0.1 + cb[None, :, None, None], 1e-5, "unary")
    close(E.map_to_nchw(ops.unary(am, clamp=(0.0, 1.0))), a.clamp(0, 1), 0, "clamp")
    bv = rnd(2, 24, seed=5)
    close(E.map_to_nchw(ops.scale_add(am, bm, avec=cs.to(DEV), bvec=bv.to(DEV), alpha=0.5, beta=-2.0)),
          0.5 * a * cs[None, :, None, None] - 2.0 * b * bv[:, :, None, None], 1e-5, "scale_add")
    gate = rnd(2, 1, 7, 9, seed=6)
    close(E.map_to_nchw(ops.mul_add(am, E.nchw_to_map(gate, DEV), row_broadcast=True, c=bm, alpha=3.0, gamma=0.5)),
          3.0 * a * gate + 0.5 * b, 1e-5, "mul_add row")
    close(E.map_to_nchw(ops.mul_add(am[..., :12], am[..., 12:])), a[:, :12] * a[:, 12:], 1e-6, "simple gate")
    close(ops.colmean(am).cpu(), a.mean((2, 3)), 1e-5, "colmean")
    big = rnd(1, 180, 70, 90, seed=7)
    close(ops.colmean(E.nchw_to_map(big, DEV)).cpu(), big.mean((2, 3)), 1e-5, "colmean big")


@pytest.mark.parametrize("C,kh,kw,H,W", [(360, 3, 3, 19, 23), (360, 3, 3, 70, 91), (100, 3, 3, 64, 64), (64, 5, 5, 19, 23),
                                         (128, 1, 21, 19, 23), (64, 21, 1, 19, 23), (3, 5, 5, 19, 23)])
def test_dwconv(ops, E, C, kh, kw, H, W):
    x, w, b = rnd(2, C, H, W, seed=1), rnd(C, 1, kh, kw, seed=2), rnd(C, seed=3)
    want = F.silu(F.conv2d(x, w, b, padding=(kh // 2, kw // 2), groups=C))
    got = ops.dwconv2d(E.nchw_to_map(x, DEV), ops.pack_dwconv(w, b, DEV), act=5)
    close(E.map_to_nchw(got), want, 1e-5, "dwconv")


def test_dw3x3_strip_kernels(ops, E):
    """the register-window strip kernels (>= 65536 pixels): odd sizes (rows % 4 != 0, a partial last strip, channels % 64 != 0),
    a channel-slice input (MambaIR's x half of xz) and the gated + pooled form on NAFNet's widest level"""
    C = 72
    x, w, b = rnd(1, 2 * C, 262, 301, seed=4), rnd(C, 1, 3, 3, seed=5), rnd(C, seed=6)
    want = F.silu(F.conv2d(x[:, :C], w, b, padding=1, groups=C))
    got = ops.dwconv2d(E.nchw_to_map(x, DEV)[..., :C], ops.pack_dwconv(w, b, DEV), act=5)
    close(E.map_to_nchw(got), want, 1e-5, "dwconv strip")
    c = 64
    x, w, b = rnd(2, 2 * c, 271, 250, seed=7), rnd(2 * c, 1, 3, 3, seed=8), rnd(2 * c, seed=9)
    t = F.conv2d(x, w, b, padding=1, groups=2 * c)
    want = t[:, :c] * t[:, c:]
    g, pooled = ops.dw3x3_gate_pool(E.nchw_to_map(x, DEV), ops.pack_dwconv(w, b, DEV))
    close(E.map_to_nchw(g), want, 1e-5, "gate strip")
    close(pooled.cpu(), want.mean((2, 3)), 1e-5, "pool strip")


def test_dw3x3_gate_pool(ops, E):
    c = 96
    x, w, b = rnd(2, 2 * c, 21, 17, seed=1), rnd(2 * c, 1, 3, 3, seed=2), rnd(2 * c, seed=3)
    t = F.conv2d(x, w, b, padding=1, groups=2 * c)
    want = t[:, :c] * t[:, c:]
    g, pooled = ops.dw3x3_gate_pool(E.nchw_to_map(x, DEV), ops.pack_dwconv(w, b, DEV))
    close(E.map_to_nchw(g), want, 1e-5, "gate")
    close(pooled.cpu(), want.mean((2, 3)), 1e-5, "pool")


def test_resamplers(ops, E):
    x = rnd(2, 12, 17, 23, seed=1)
    xm = E.nchw_to_map(x, DEV)
    for size in ((68, 92), (34, 46), (8, 11), (17, 23), (35, 12)):
        close(E.map_to_nchw(ops.bilinear(xm, *size)), F.interpolate(x, size=size, mode="bilinear", align_corners=False),
              1e-5, f"bilinear {size}")
    y = rnd(1, 3, 13, 15, seed=2).abs()
    close(E.map_to_nchw(ops.bicubic_up(E.nchw_to_map(y, DEV), 4)),
          F.interpolate(y, scale_factor=4, mode="bicubic", align_corners=False), 1e-5, "bicubic")
    close(E.map_to_nchw(ops.avgpool2(xm)), F.avg_pool2d(x, 2, 2), 1e-6, "avgpool")
    z = rnd(1, 3, 20, 27, seed=3)
    zm = E.nchw_to_map(z, DEV)
    close(E.map_to_nchw(ops.pad_reflect(zm, 32, 32)), F.pad(z, (0, 5, 0, 12), mode="reflect"), 0, "pad16")
    close(E.map_to_nchw(ops.crop(zm, 11, 9, clamp=True)), z[:, :, :11, :9].clamp(0, 1), 0, "crop")


def test_uint8_boundary(ops):
    img = torch.randint(0, 256, (1, 9, 11, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    m = ops.u8_to_map(img.to(DEV))
    assert torch.equal(m.cpu(), img.float().div(255.0))
    v = torch.tensor([0.5, 1.5, 2.5, 254.5, 300.0, -3.0]).div(255.0).reshape(1, 1, 2, 3)
    vm = ops.new_map(1, 1, 2, 3, DEV)
    vm.copy_(v)
    assert ops.map_to_u8(vm).cpu().flatten().tolist() == [0, 2, 2, 254, 255, 0]


@pytest.mark.parametrize("variant,tol", [(0, 1e-4), (4, 1e-4), (3, 2e-5)])   # split-bf16 MFMA kernel (512 / 256 threads), exact f32-MFMA kernel
@pytest.mark.parametrize("C,heads,shift,H,W", [(180, 6, 0, 32, 48), (212, 4, 8, 32, 32), (244, 2, 0, 16, 32),
                                                (276, 6, 8, 48, 32), (308, 4, 0, 32, 32), (60, 6, 8, 32, 32)])
def test_window_attn(ops, C, heads, shift, H, W, variant, tol):
    from ffsr_oracle.common import win_split, win_merge, shift_mask
    B, ws, hd = 2, 16, C // heads
    from ffsr_oracle.drct import rel_pos_index
    qkv = rnd(B * H * W, 3 * C, seed=1)
    table = rnd(31 * 31, heads, seed=2)
    bias = table[rel_pos_index(16).reshape(-1)].reshape(256, 256, heads).permute(2, 0, 1)     # [heads, q, k]
    t = qkv.reshape(B, H, W, 3 * C)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    w = win_split(t, ws, ws).reshape(-1, 256, 3, heads, hd).permute(2, 0, 3, 1, 4)
    a = (w[0] * hd ** -0.5) @ w[1].transpose(-2, -1) + bias[None]
    if shift:
        m = shift_mask(H, W, ws, ws, shift, shift)
        a = (a.reshape(B, -1, heads, 256, 256) + m[None, :, None]).reshape(-1, heads, 256, 256)
    o = win_merge((a.softmax(-1) @ w[2]).transpose(1, 2).reshape(-1, 256, C), ws, ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    got = ops.window_attn(qkv.to(DEV), table.to(DEV), B, H, W, C, heads, ws, shift, hd ** -0.5, variant=variant)
    close(got.cpu(), o.reshape(B * H * W, C), tol, "window attention")


@pytest.mark.parametrize("heads,hd,shift,H,W", [(3, 30, 0, 16, 24), (3, 30, 4, 24, 16), (2, 10, 4, 16, 16)])
def test_grl_window_attn(ops, heads, hd, shift, H, W):
    """GRL's cosine window attention (mixed_attn_block_efficient.py:77-165): normalised q.k * logit + CPB bias (+ shift
    mask), one branch of the shared qkv tensor (column offset, [3][heads][hd] layout)."""
    from ffsr_oracle.common import win_split, win_merge, shift_mask
    B, ws, Cb = 2, 8, heads * hd
    col0, ldq = 6, 3 * Cb + 10
    qkv = rnd(B * H * W, ldq, seed=1)
    bias = rnd(heads, 64, 64, seed=2)                        # [heads, query, key]
    logit = rnd(heads, seed=3).abs() + 0.5
    t = qkv[:, col0:col0 + 3 * Cb].reshape(B, H, W, 3 * Cb)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    w = win_split(t, ws, ws).reshape(-1, 64, 3, heads, hd).permute(2, 0, 3, 1, 4)
    a = F.normalize(w[0], dim=-1) @ F.normalize(w[1], dim=-1).transpose(-2, -1) * logit[None, :, None, None] + bias[None]
    if shift:
        m = shift_mask(H, W, ws, ws, shift, shift)
        a = (a.reshape(B, -1, heads, 64, 64) + m[None, :, None]).reshape(-1, heads, 64, 64)
    o = win_merge((a.softmax(-1) @ w[2]).transpose(1, 2).reshape(-1, 64, Cb), ws, ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    out = torch.zeros(B * H * W, Cb + 8, device=DEV)
    ops.grl_window_attn(qkv.to(DEV), col0, bias.permute(0, 2, 1).contiguous().to(DEV), logit.to(DEV), out, 4, B, H, W, heads,
                        hd, shift)
    close(out[:, 4:4 + Cb].cpu(), o.reshape(B * H * W, Cb), 2e-5, "grl window attention")
    assert (out[:, :4] == 0).all() and (out[:, 4 + Cb:] == 0).all()


def test_pixel_mha(ops):
    for T, E_, heads in ((9, 64, 4), (4, 128, 8)):
        S = 301
        qkv = rnd(S * T, 3 * E_, seed=T)
        q, k, v = (qkv.reshape(S, T, 3, heads, 16)[:, :, i].transpose(1, 2) for i in range(3))
        want = ((q / 4) @ k.transpose(-2, -1)).softmax(-1) @ v
        got = ops.pixel_mha(qkv.to(DEV), S, T, E_, heads)
        close(got.cpu(), want.transpose(1, 2).reshape(S * T, E_), 1e-5, "pixel mha")


@pytest.mark.parametrize("H,W,Dm,R,chunk", [(8, 12, 96, 3, 32), (16, 16, 360, 12, 64), (7, 9, 360, 12, None)])
def test_selective_scan(ops, H, W, Dm, R, chunk):
    from ffsr_oracle.scan import selective_scan_ref
    B, L, N = 2, H * W, 16
    u, xdbl = rnd(B, L, Dm, seed=1), rnd(B, L, 4 * (R + 32), seed=2)
    dtw, dtb = rnd(4, Dm, R, seed=3, scale=0.3), rnd(4, Dm, seed=4)
    A, Dv = -(torch.rand(4 * Dm, N, generator=torch.Generator().manual_seed(5)) * 2 + 0.05), rnd(4 * Dm, seed=6)
    # oracle formulation: gathered directions, [B, 4*Dm, L]
    um = u.transpose(1, 2).reshape(B, Dm, H, W)
    rows, cols = um.reshape(B, Dm, L), um.transpose(2, 3).reshape(B, Dm, L)
    xs = torch.stack([rows, cols, rows.flip(-1), cols.flip(-1)], 1)
    xm = xdbl.transpose(1, 2).reshape(B, 4, R + 32, H, W)
    def order(t, k):
        t = t.reshape(B, -1, L) if k % 2 == 0 else t.transpose(2, 3).reshape(B, -1, L)
        return t.flip(-1) if k >= 2 else t
    proj = torch.stack([order(xm[:, k], k) for k in range(4)], 1)          # [B,4,R+32,L]
    dts = torch.einsum("bkrl,kdr->bkdl", proj[:, :, :R], dtw)
    y = selective_scan_ref(xs.reshape(B, 4 * Dm, L), dts.reshape(B, 4 * Dm, L), A, proj[:, :, R:R + N].contiguous(),
                           proj[:, :, R + N:].contiguous(), Dv, delta_bias=dtb.reshape(-1), delta_softplus=True)
    y = y.reshape(B, 4, Dm, L)
    def unorder(t, k):
        t = t.flip(-1) if k >= 2 else t
        return t if k % 2 == 0 else t.reshape(B, Dm, W, H).transpose(2, 3).reshape(B, Dm, L)
    want = torch.stack([unorder(y[:, k], k) for k in range(4)], 0).transpose(2, 3).reshape(4, B * L, Dm)
    got = ops.selective_scan4(u.reshape(B * L, Dm).to(DEV), xdbl.reshape(B * L, -1).to(DEV), dtw.to(DEV), dtb.to(DEV),
                              A.to(DEV), Dv.to(DEV), B, H, W, Dm, R, chunk=chunk, pairs=False)
    close(got.cpu(), want, 5e-5, "selective scan")
    z, g, b = rnd(B * L, Dm, seed=7), rnd(Dm, seed=8), rnd(Dm, seed=9)
    gated = ops.mamba_norm_gate(got, z.to(DEV), g.to(DEV), b.to(DEV))
    close(gated.cpu(), F.layer_norm(want.sum(0), (Dm,), g, b) * F.silu(z), 5e-5, "norm gate")
    # pair planes: the directions 2 / 3 are scanned after 0 / 1 and add to their planes
    two = ops.selective_scan4(u.reshape(B * L, Dm).to(DEV), xdbl.reshape(B * L, -1).to(DEV), dtw.to(DEV), dtb.to(DEV),
                              A.to(DEV), Dv.to(DEV), B, H, W, Dm, R, chunk=chunk, pairs=True)
    assert tuple(two.shape) == (2, B * L, Dm)
    close(two.cpu(), torch.stack([want[0] + want[2], want[1] + want[3]]), 1e-4, "selective scan, pair planes")
    gated2 = ops.mamba_norm_gate(two, z.to(DEV), g.to(DEV), b.to(DEV))
    close(gated2.cpu(), F.layer_norm(want.sum(0), (Dm,), g, b) * F.silu(z), 5e-5, "norm gate over pair planes")


@pytest.mark.parametrize("h,w", [(64, 64), (35, 51)])
def test_frequency_bands(ops, E, pkg, h, w):
    import importlib
    from conftest import load_golden
    from ffsr_oracle import fusion as ofusion
    fus = importlib.import_module("image-super-resolution_amd.fusion")
    g = load_golden("fusion_full.pt")
    net = fus.FusionNet(g["sd"], DEV)
    lr = torch.rand(2, 3, h, w, generator=torch.Generator().manual_seed(3))
    bands = net.frequency_bands(E.nchw_to_map(lr, DEV)).cpu().reshape(2, h, w, 9, 4)
    want = ofusion.frequency_bands(g["sd"], lr)
    for i, b in enumerate(want):
        close(bands[..., i, :3].permute(0, 3, 1, 2), b, 2e-5, f"band {i}")
    assert bands[..., 3].abs().max() == 0


@pytest.mark.parametrize("B,H,W,C,sq", [(1, 352, 512, 180, 6), (2, 17, 23, 180, 10), (3, 8, 8, 60, 3)])
def test_channel_attention_tail(ops, E, B, H, W, C, sq):
    """RCAN channel attention of CAB (mambair_arch.py:20-38): AdaptiveAvgPool2d(1) -> 1x1 conv -> ReLU -> 1x1 conv -> sigmoid
    in two launches, against torch"""
    g = torch.Generator().manual_seed(C + sq)
    x = torch.randn(B, C, H, W, generator=g)
    w1, b1 = torch.randn(sq, C, 1, 1, generator=g) * 0.2, torch.randn(sq, generator=g)
    w2, b2 = torch.randn(C, sq, 1, 1, generator=g), torch.randn(C, generator=g)
    want = torch.sigmoid(F.conv2d(F.relu(F.conv2d(x.mean((2, 3), keepdim=True), w1, b1)), w2, b2))[:, :, 0, 0]
    a1 = ops.pack_conv(w1, b1, DEV)
    a3 = ops.pack_conv(w2, b2, DEV, cin_pad=ops.pad4(sq))
    got = ops.channel_attention(E.nchw_to_map(x, DEV), a1, a3).cpu()
    assert tuple(got.shape) == (B, C) and (got - want).abs().max().item() < 2e-6


@pytest.mark.parametrize("B,H,W,Cin,N,act,res", [(1, 96, 160, 64, 3, 0, False), (2, 67, 131, 128, 3, 0, True),
                                                 (1, 70, 70, 16, 4, 1, False), (1, 64, 80, 8, 1, 4, False),
                                                 (1, 65, 64, 32, 2, 3, True), (3, 40, 37, 8, 2, 2, False)])
def test_thin_conv3x3_heads(ops, B, H, W, Cin, N, act, res):
    """The N <= 4 3x3 heads at HR resolution (conv_last drct_arch.py:789, refine.10 enhanced_fusion_v2.py:576, the edge gates
    edge_enhancement.py:88,177 ...) on ffsr_conv3x3_thin_f32: exact fp32 FMA against an fp64 torch convolution, image borders,
    partial row blocks and ragged runs included; the dispatcher takes it for these shapes and only these."""
    g = torch.Generator().manual_seed(Cin * 10 + N)
    x = torch.randn(B, Cin, H, W, generator=g)
    w, b = torch.randn(N, Cin, 3, 3, generator=g) * 0.1, torch.randn(N, generator=g)
    r = torch.randn(B, N, H, W, generator=g) if res else None
    f = {0: lambda t: t, 1: F.gelu, 2: F.relu, 3: lambda t: F.leaky_relu(t, 0.2), 4: torch.sigmoid}[act]
    want = f(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    if res:
        want = want * 0.1 + r.double()
    cv = ops.pack_conv(w, b, DEV)
    assert ops.thin3_ok(cv, B * H * W)
    xm = torch.empty(B, H, W, Cin, device=DEV).copy_(x.permute(0, 2, 3, 1))
    rm = None
    if res:
        rm = ops.new_map(B, H, W, N, DEV)
        rm.copy_(r.permute(0, 2, 3, 1))
    got = ops.conv2d(xm, cv, act=act, slope=0.2 if act == 3 else 0.0, res=rm, cscale=0.1 if res else 1.0)
    assert (got.permute(0, 3, 1, 2).double().cpu() - want).abs().max().item() < 5e-6
    # not eligible: more output channels than the pixel has lanes, or a channel count the kernel has no instance for
    assert not ops.thin3_ok(ops.pack_conv(torch.randn(3, 8, 3, 3), None, DEV), 1 << 20)
    assert not ops.thin3_ok(ops.pack_conv(torch.randn(3, 48, 3, 3), None, DEV), 1 << 20)
    assert not ops.thin3_ok(ops.pack_conv(torch.randn(8, 64, 3, 3), None, DEV), 1 << 20)
