"""ffsr_tok_chain_f32 (token-stationary fused chains) against float64 torch: the Swin Mlp with its LayerNorm and residual
(drct_arch.py:77-95, :405-407), GRL's post-norm form (mixed_attn_block_efficient.py:543-554) and NAFBlock's gated second half
(nafnet_arch.py:125-131).  Tolerance: the split-bf16 products carry ~1e-5 relative error per GEMM (north_star: 1e-3)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def mod(name):
    return importlib.import_module("image-super-resolution_amd." + name)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def rel(got, want):
    return (got.double().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-20)


@pytest.mark.parametrize("K,H,M,waves", [(180, 360, 1000, 8), (180, 360, 5000, 4), (212, 424, 777, 8), (244, 488, 2048, 4),
                                         (276, 276, 1500, 8), (308, 308, 4096 + 16, 8), (180, 360, 352 * 512 + 40, 8)])
def test_swin_mlp_prenorm_residual(K, H, M, waves, monkeypatch):
    """x + fc2(GELU(fc1(LayerNorm(x)))) on a channel-slice view of a wider buffer (DRCT's dense-concat buffer)."""
    ops = mod("ops")
    monkeypatch.setattr(ops, "TOK_WAVES", waves)
    g = gen(K + M)
    wide = torch.randn(M, 308, generator=g) * 1.5 + 0.3
    x = wide[:, :K]
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    w1, b1 = torch.randn(H, K, generator=g) / K ** 0.5, torch.randn(H, generator=g) * 0.1
    w2, b2 = torch.randn(K, H, generator=g) / H ** 0.5, torch.randn(K, generator=g) * 0.1
    xd = x.double()
    n = F.layer_norm(xd, (K,), gamma.double(), beta.double(), 1e-5)
    want = xd + F.linear(F.gelu(F.linear(n, w1.double(), b1.double())), w2.double(), b2.double())
    tc = ops.pack_tok_chain(w1, b1, w2, b2, DEV, mode=0, ln=(gamma, beta), eps=1e-5)
    xg = wide.to(DEV)[:, :K]
    got = ops.tok_chain(xg, tc, res=xg)
    assert tuple(got.shape) == (M, K)
    assert rel(got, want) < 3e-5
    # the output may also land in a channel slice of a wider buffer
    buf = torch.full((M, 308), 7.0, device=DEV)
    ops.tok_chain(xg, tc, res=xg, out=buf[:, :K])
    assert torch.equal(buf[:, :K], got) and bool((buf[:, K:] == 7.0).all())


@pytest.mark.parametrize("M", [640, 1027])
def test_grl_mlp_postnorm_planes(M):
    """y + LayerNorm(fc2(GELU(fc1(y)))) with fp32 and bf16-plane outputs (hi + lo == fp32 to 2^-17, pad columns zero)."""
    ops = mod("ops")
    K, H = 180, 360
    g = gen(M)
    y = torch.randn(M, K, generator=g)
    w1, b1 = torch.randn(H, K, generator=g) / K ** 0.5, torch.randn(H, generator=g) * 0.1
    w2, b2 = torch.randn(K, H, generator=g) / H ** 0.5, torch.randn(K, generator=g) * 0.1
    g2, be2 = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    yd = y.double()
    m = F.linear(F.gelu(F.linear(yd, w1.double(), b1.double())), w2.double(), b2.double())
    want = yd + F.layer_norm(m, (K,), g2.double(), be2.double(), 1e-5)
    tc = ops.pack_tok_chain(w1, b1, w2, b2, DEV, mode=0)
    yg = y.to(DEV)
    got, pl = ops.tok_chain(yg, tc, post_ln=(g2.to(DEV), be2.to(DEV)), res2=yg, out_planes=True)
    assert rel(got, want) < 3e-5
    back = pl.buf[0].float() + pl.buf[1].float()
    assert pl.Cp == 192 and bool((back[:, K:] == 0).all())
    assert (back[:, :K] - got).abs().max().item() <= 2.0 ** -16 * got.abs().max().item()
    only = ops.tok_chain(yg, tc, post_ln=(g2.to(DEV), be2.to(DEV)), res2=yg, out_planes=True, want_f32=False)
    assert torch.equal(only.buf, pl.buf)


@pytest.mark.parametrize("c,M", [(64, 3000), (128, 1111)])
def test_nafnet_gated_half(c, M):
    """y + gamma * conv5(SimpleGate(conv4(LayerNorm2d(y)))) per pixel (nafnet_arch.py:125-131; eps 1e-6)."""
    ops = mod("ops")
    g = gen(c + M)
    y = torch.randn(M, c, generator=g) * 2.0
    ln_w, ln_b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    w4, b4 = torch.randn(2 * c, c, generator=g) / c ** 0.5, torch.randn(2 * c, generator=g) * 0.1
    w5, b5 = torch.randn(c, c, generator=g) / c ** 0.5, torch.randn(c, generator=g) * 0.1
    gam = torch.randn(c, generator=g) * 0.3
    yd = y.double()
    t = F.linear(F.layer_norm(yd, (c,), ln_w.double(), ln_b.double(), 1e-6), w4.double(), b4.double())
    want = yd + F.linear(t[:, :c] * t[:, c:], w5.double(), b5.double()) * gam.double()
    tc = ops.pack_tok_chain(w4, b4, w5, b5, DEV, mode=1, ln=(ln_w, ln_b), eps=1e-6)
    yg = y.to(DEV)
    got = ops.tok_chain(yg, tc, res=yg, cvec=gam.to(DEV))
    assert rel(got, want) < 5e-5


def test_gelu_fast_matches_erf_gelu():
    """the kernel's GELU (erfc by Abramowitz-Stegun 7.1.26) against torch's exact-erf GELU through an identity chain:
    W1 = I (K = 64 -> hidden 64 via the gate-free mode is not available, so use mode 0 with K = N = 180 and W2 = I)."""
    ops = mod("ops")
    K = 180
    x = torch.linspace(-9.0, 9.0, 1024 * K).reshape(1024, K)
    eye = torch.eye(K)
    tc = ops.pack_tok_chain(eye, None, eye, None, DEV, mode=0)
    got = ops.tok_chain(x.to(DEV), tc).cpu()
    want = F.gelu(x.double())
    # identity weights are exact in bf16; the input and the hidden value are carried as bf16 hi + lo (2^-17 relative each);
    # the approximation itself adds <= 1.5e-7 * |x| / 2
    err = (got.double() - want).abs()
    assert (err - 2.0 ** -15 * want.abs()).max().item() < 1e-6
    assert err[(x.abs() < 1.0)].max().item() < 1e-5


@pytest.mark.parametrize("K,N,M,ln,bias,act", [(180, 540, 1000, True, True, 0), (308, 924, 2051, True, True, 0), (180, 720, 4099, True, False, 0),
                                               (360, 176, 2500, False, False, 0), (360, 176, 352 * 512 + 8, False, False, 0),
                                               (64, 128, 3333, True, True, 0), (128, 256, 700, True, True, 0), (244, 32, 1500, False, True, 3),
                                               (212, 636, 352 * 512 + 8, True, True, 0), (276, 180, 900, False, True, 1)])
def test_tok_gemm_with_layernorm_prologue(K, N, M, ln, bias, act):
    """ffsr_tok_gemm_f32: act(W LayerNorm(x) + b) -- norm1 + qkv (drct_arch.py:385-388, :166), ln_1 + in_proj
    (mambair_arch.py:417, :238), norm1 + conv1 (nafnet_arch.py:113-115) and the LeakyReLU(0.2) adjust convolutions writing a
    channel slice of the dense-concat buffer (drct_arch.py:292-301)."""
    ops = mod("ops")
    g = gen(K * N + M)
    wide = torch.randn(M, max(308, K + 24), generator=g) * 1.5 + 0.2
    x = wide[:, :K]
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1 if bias else None
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    xd = x.double()
    n = F.layer_norm(xd, (K,), gamma.double(), beta.double(), 1e-5) if ln else xd
    want = F.linear(n, w.double(), None if b is None else b.double())
    if act == 3:
        want = F.leaky_relu(want, 0.2)
    elif act == 1:
        want = F.gelu(want)
    tg = ops.pack_tok_gemm(w, b, DEV, ln=(gamma, beta) if ln else None)
    xg = wide.to(DEV)[:, :K]
    if N == 32:       # into a slice of a wider buffer, the rest untouched
        buf = torch.full((M, 308), 7.0, device=DEV)
        ops.tok_gemm(xg, tg, act=act, slope=0.2, out=buf[:, K:K + N])
        got = buf[:, K:K + N]
        assert bool((buf[:, :K] == 7.0).all()) and bool((buf[:, K + N:] == 7.0).all())
    else:
        got, pl = ops.tok_gemm(xg, tg, act=act, slope=0.2, out_planes=True)
        back = pl.buf[0].float() + pl.buf[1].float()
        assert bool((back[:, N:] == 0).all())
        assert (back[:, :N] - got).abs().max().item() <= 2.0 ** -16 * got.abs().max().item()
    assert rel(got, want) < 3e-5


@pytest.mark.parametrize("K,H,N3,M,act", [(180, 360, 32, 1000, 3), (244, 488, 32, 2051, 3), (308, 308, 180, 1500, 0), (276, 276, 32, 352 * 512 + 8, 3)])
def test_mlp_with_adjust_tail(K, H, N3, M, act):
    """the Swin MLP kernel with the dense block's adjust convolution as its tail (drct_arch.py:292-301): adjust1-4 + LeakyReLU(0.2)
    write a 32-channel slice of the concatenation buffer, adjust5 * 0.2 + x the next group's input; the MLP output itself is only
    stored on request"""
    ops = mod("ops")
    g = gen(K + M + N3)
    wide = torch.randn(M, 308, generator=g) * 1.5 + 0.3
    x = wide[:, :K]
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    w1, b1 = torch.randn(H, K, generator=g) / K ** 0.5, torch.randn(H, generator=g) * 0.1
    w2, b2 = torch.randn(K, H, generator=g) / H ** 0.5, torch.randn(K, generator=g) * 0.1
    w3, b3 = torch.randn(N3, K, generator=g) / K ** 0.5, torch.randn(N3, generator=g) * 0.1
    xd = x.double()
    y = xd + F.linear(F.gelu(F.linear(F.layer_norm(xd, (K,), gamma.double(), beta.double(), 1e-5), w1.double(), b1.double())), w2.double(), b2.double())
    z = F.linear(y, w3.double(), b3.double())
    want3 = F.leaky_relu(z, 0.2) if act == 3 else z * 0.2 + wide[:, :N3].double()
    tc = ops.pack_tok_chain(w1, b1, w2, b2, DEV, mode=0, ln=(gamma, beta))
    tg = ops.pack_tok_gemm(w3, b3, DEV)
    wg = wide.to(DEV)
    xg = wg[:, :K]
    if act == 3:        # into a slice right of the input columns, everything else untouched
        before = wg.clone()
        out3 = wg[:, K:K + N3] if K + N3 <= 308 else torch.empty(M, N3, device=DEV)
        r = ops.tok_chain(xg, tc, res=xg, tail=dict(tg=tg, out=out3, act=3, slope=0.2), want_f32=False)
        assert r is out3
        if K + N3 <= 308:
            assert torch.equal(wg[:, :K], before[:, :K]) and torch.equal(wg[:, K + N3:], before[:, K + N3:])
        assert rel(out3, want3) < 3e-5
    else:
        out3 = torch.empty(M, N3, device=DEV)
        ygot, r = ops.tok_chain(xg, tc, res=xg, tail=dict(tg=tg, out=out3, cscale=0.2, res=wg[:, :N3]))
        assert rel(ygot, y) < 3e-5 and rel(r, want3) < 3e-5


@pytest.mark.parametrize("K,H,M,waves,tail", [(180, 360, 1000, 8, False), (180, 360, 3000, 4, True), (212, 424, 777, 8, True),
                                              (244, 488, 2048, 8, False), (276, 276, 1500, 8, True), (308, 308, 4096 + 16, 8, True),
                                              (180, 360, 352 * 512 + 40, 8, True)])
def test_swin_block_head_proj_mlp(K, H, M, waves, tail, monkeypatch):
    """the attention's output projection as the HEAD of the MLP kernel (drct_arch.py:400-407): x1 = x + proj(a);
    y = x1 + fc2(GELU(fc1(norm2(x1)))) [; adjust tail on y].  x is a channel slice of the dense-concat buffer; x1 is never stored."""
    ops = mod("ops")
    monkeypatch.setattr(ops, "TOK_WAVES", waves)
    g = gen(K + M + 1)
    wide = torch.randn(M, 308, generator=g) * 1.5 + 0.3
    x = wide[:, :K]
    a = torch.randn(M, K, generator=g)
    w0, b0 = torch.randn(K, K, generator=g) / K ** 0.5, torch.randn(K, generator=g) * 0.1
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    w1, b1 = torch.randn(H, K, generator=g) / K ** 0.5, torch.randn(H, generator=g) * 0.1
    w2, b2 = torch.randn(K, H, generator=g) / H ** 0.5, torch.randn(K, generator=g) * 0.1
    w3, b3 = torch.randn(32, K, generator=g) / K ** 0.5, torch.randn(32, generator=g) * 0.1
    x1 = x.double() + F.linear(a.double(), w0.double(), b0.double())
    y = x1 + F.linear(F.gelu(F.linear(F.layer_norm(x1, (K,), gamma.double(), beta.double(), 1e-5), w1.double(), b1.double())),
                      w2.double(), b2.double())
    head = ops.pack_tok_gemm(w0, b0, DEV)
    tc = ops.pack_tok_chain(w1, b1, w2, b2, DEV, mode=0, ln=(gamma, beta))
    wg, ag = wide.to(DEV), a.to(DEV)
    if not tail:
        got = ops.tok_head_chain(ag, head, tc, hres=wg[:, :K])
        assert rel(got, y) < 4e-5
        return
    want3 = F.leaky_relu(F.linear(y, w3.double(), b3.double()), 0.2)
    tg = ops.pack_tok_gemm(w3, b3, DEV)
    out3 = torch.empty(M, 32, device=DEV)
    ygot, r = ops.tok_head_chain(ag, head, tc, hres=wg[:, :K], tail=dict(tg=tg, out=out3, act=3, slope=0.2))
    assert r is out3 and rel(ygot, y) < 4e-5 and rel(out3, want3) < 4e-5
    only = torch.empty(M, 32, device=DEV)
    assert ops.tok_head_chain(ag, head, tc, hres=wg[:, :K], tail=dict(tg=tg, out=only, act=3, slope=0.2), want_f32=False) is only
    assert torch.equal(only, out3)


@pytest.mark.parametrize("M,per", [(640, 320), (2 * 1027, 1027), (352 * 512, 352 * 512)])
def test_grl_block_head_proj_norm_mlp(M, per):
    """GRL's block tail as one kernel (mixed_attn_block_efficient.py:536-554): y = x + norm1(proj(a)) + c2 * att[image];
    out = y + norm2(fc2(GELU(fc1(y)))), fp32 and bf16 planes; y is never stored."""
    ops = mod("ops")
    K, H = 180, 360
    g = gen(M)
    x, a, c2 = (torch.randn(M, K, generator=g) for _ in range(3))
    att = torch.rand(M // per, K, generator=g)
    w0, b0 = torch.randn(K, K, generator=g) / K ** 0.5, torch.randn(K, generator=g) * 0.1
    g0, be0 = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    w1, b1 = torch.randn(H, K, generator=g) / K ** 0.5, torch.randn(H, generator=g) * 0.1
    w2, b2 = torch.randn(K, H, generator=g) / H ** 0.5, torch.randn(K, generator=g) * 0.1
    g2, be2 = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    y = x.double() + F.layer_norm(F.linear(a.double(), w0.double(), b0.double()), (K,), g0.double(), be0.double(), 1e-5) \
        + c2.double() * att.double().repeat_interleave(per, 0)
    m = F.linear(F.gelu(F.linear(y, w1.double(), b1.double())), w2.double(), b2.double())
    want = y + F.layer_norm(m, (K,), g2.double(), be2.double(), 1e-5)
    head = ops.pack_tok_gemm(w0, b0, DEV)
    tc = ops.pack_tok_chain(w1, b1, w2, b2, DEV, mode=0)
    d = lambda t: t.to(DEV)
    got, pl = ops.tok_head_chain(d(a), head, tc, head_ln=(d(g0), d(be0)), hres=d(x), hres2=d(c2), hvec2=d(att), rows_per_batch=per,
                                 post_ln=(d(g2), d(be2)), out_planes=True)
    assert rel(got, want) < 4e-5
    back = pl.buf[0].float() + pl.buf[1].float()
    assert pl.Cp == 192 and bool((back[:, K:] == 0).all())
    assert (back[:, :K] - got).abs().max().item() <= 2.0 ** -16 * got.abs().max().item()


@pytest.mark.parametrize("M,full", [(1000, True), (2051, False), (352 * 512 + 24, True)])
def test_mamba_out_proj_with_gate_prologue_and_ln2(M, full):
    """MambaIR's VSS tail as one kernel (mambair_arch.py:381-386, :417-419): a = out_norm(y0 + y1 + y2 + y3) * silu(z);
    y = x * skip_scale + out_proj(a); planes <- ln_2(y), fp32 <- y.  `full` False: the plain projection (no prologue, no post-LN)."""
    ops = mod("ops")
    K, N = 360, 180
    g = gen(M)
    y4 = torch.randn(4, M, K, generator=g)
    xz = torch.randn(M, 2 * K, generator=g)
    x = torch.randn(M, N, generator=g)
    w0 = torch.randn(N, K, generator=g) / K ** 0.5
    pg, pb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    g2, be2 = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    skip = torch.rand(N, generator=g) + 0.5
    tg = ops.pack_tok_gemm(w0, None, DEV, check=False)
    d = lambda t: t.to(DEV)
    if not full:
        want = F.linear(y4[0].double(), w0.double()) * 0.5 + x.double()
        got = ops.tok_proj(d(y4)[0], tg, cscale=0.5, res=d(x))
        assert rel(got, want) < 4e-5
        return
    z = xz[:, K:].double()
    a = F.layer_norm(y4.double().sum(0), (K,), pg.double(), pb.double(), 1e-5) * F.silu(z)
    y = x.double() * skip.double() + F.linear(a, w0.double())
    n = F.layer_norm(y, (N,), g2.double(), be2.double(), 1e-5)
    y4g, xzg = d(y4), d(xz)
    got, pl = ops.tok_proj(y4g[0], tg, xdirs=4, xstride=y4g.stride(0), z=xzg[:, K:], pro_ln=(d(pg), d(pb)), res=d(x), rvec=d(skip),
                           post_ln=(d(g2), d(be2)), out_pre_ln=True, out_planes=True)
    assert rel(got, y) < 4e-5
    back = pl.buf[0].float() + pl.buf[1].float()
    assert pl.Cp == 192 and bool((back[:, N:] == 0).all())
    assert rel(back[:, :N], n) < 4e-5


@pytest.mark.parametrize("c,M,per", [(64, 3000, 1500), (128, 1111, 1111), (64, 1024 * 96 + 8, 1024 * 96 + 8)])
def test_nafnet_block_head_conv3_sca_gated_half(c, M, per):
    """NAFBlock from the gated map on, one kernel (nafnet_arch.py:122-131): y = x + beta * conv3(g * sca[image]);
    out = y + gamma * conv5(SimpleGate(conv4(LayerNorm2d(y))))  (eps 1e-6); y is never stored."""
    ops = mod("ops")
    gn = gen(c + M)
    x, g = torch.randn(M, c, generator=gn), torch.randn(M, c, generator=gn)
    sca = torch.rand(M // per, c, generator=gn) + 0.5
    w3, b3 = torch.randn(c, c, generator=gn) / c ** 0.5, torch.randn(c, generator=gn) * 0.1
    beta, gamma = torch.randn(c, generator=gn) * 0.5, torch.randn(c, generator=gn) * 0.5
    n2w, n2b = torch.rand(c, generator=gn) + 0.5, torch.randn(c, generator=gn) * 0.1
    w4, b4 = torch.randn(2 * c, c, generator=gn) / c ** 0.5, torch.randn(2 * c, generator=gn) * 0.1
    w5, b5 = torch.randn(c, c, generator=gn) / c ** 0.5, torch.randn(c, generator=gn) * 0.1
    y = x.double() + beta.double() * F.linear(g.double() * sca.double().repeat_interleave(per, 0), w3.double(), b3.double())
    t = F.linear(F.layer_norm(y, (c,), n2w.double(), n2b.double(), 1e-6), w4.double(), b4.double())
    want = y + gamma.double() * F.linear(t[:, :c] * t[:, c:], w5.double(), b5.double())
    head = ops.pack_tok_gemm(w3 * beta[:, None], b3 * beta, DEV)
    tc = ops.pack_tok_chain(w4, b4, w5 * gamma[:, None], b5 * gamma, DEV, mode=1, ln=(n2w, n2b), eps=1e-6)
    got = ops.tok_head_chain(g.to(DEV), head, tc, in_scale=sca.to(DEV), hres=x.to(DEV), rows_per_batch=per)
    assert rel(got, want) < 4e-5
