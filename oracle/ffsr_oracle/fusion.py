"""7-phase frequency-guided fusion (CompleteEnhancedFusionSR, eval mode) -- CPU oracle.

TEST INFRASTRUCTURE.  Follows src/models/enhanced_fusion_v2.py (_run_pipeline :681,
DynamicExpertSelector.forward :450), multi_domain_frequency.py (DCT :146, DWT :273, FFT :352),
large_kernel_attention.py (LKA :92, LKABlock :143, cross-band :207, collaborative :324),
hierarchical_fusion.py :131 and edge_enhancement.py (pyramid :178, refine block :109, forward :222).
``sd`` = reference state_dict (BatchNorm in eval mode uses the running statistics).
"""
import math
import torch
import torch.nn.functional as F

EXPERTS = ("drct", "grl", "nafnet", "mamba")


def bilinear(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def conv(sd, p, x, pad=0, **kw):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), padding=pad, **kw)


def bn_eval(sd, p, x, eps=1e-5, train=False):
    """nn.BatchNorm2d: running statistics in eval mode; in train mode batch statistics, and the running statistics in
    ``sd`` are updated in place (momentum 0.1, unbiased variance), as model.train() does."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], train, 0.1 if train else 0.0, eps)


# ------------------------------------------------------------------ phase 2: 9 frequency bands
def dct_bands(sd, x, p="freq_decomp.dct."):
    B, C, H, W = x.shape
    N = sd[p + "dct_basis"].shape[0]
    ph, pw = (N - H % N) % N, (N - W % N) % N
    xp = F.pad(x, (0, pw, 0, ph), mode="reflect") if (ph or pw) else x
    Hp, Wp = xp.shape[-2:]
    blk = xp.reshape(B, C, Hp // N, N, Wp // N, N).permute(0, 1, 2, 4, 3, 5)
    D = sd[p + "dct_basis"]
    coef = D @ blk @ D.t()
    out = []
    for i, m in enumerate(("low_mask", "mid_mask", "high_mask")):
        s = D.t() @ (coef * sd[p + m]) @ D
        s = s.permute(0, 1, 2, 4, 3, 5).reshape(B, C, Hp, Wp)[:, :, :H, :W]
        out.append(s * sd[p + "band_scale"][i])
    return out


def dwt_bands(sd, x, p="freq_decomp.dwt."):
    B, C, H, W = x.shape
    lo_r, hi_r, lo_c, hi_c = (sd[p + k] for k in ("lo_row", "hi_row", "lo_col", "hi_col"))
    pad = lo_r.shape[-1] - 1
    xr = F.pad(x, (pad, pad, 0, 0), mode="reflect")
    rows = [F.conv2d(xr, f, stride=(1, 2), groups=C) for f in (lo_r, hi_r)]
    subs = []
    for r in rows:
        rc = F.pad(r, (0, 0, pad, pad), mode="reflect")
        subs += [F.conv2d(rc, f, stride=(2, 1), groups=C) for f in (lo_c, hi_c)]
    return [bilinear(s, (H, W)) * sd[p + "subband_scale"][i] for i, s in enumerate(subs)]


def fft_bands(sd, x, p="freq_decomp.fft."):
    X = torch.fft.rfft2(x, norm="ortho")
    m = bilinear(sd[p + "freq_mask_logits"], X.shape[-2:])
    m = torch.sigmoid(m * sd[p + "temperature"].clamp(min=1.0))
    lo = torch.fft.irfft2(X * m, s=x.shape[-2:], norm="ortho")
    hi = torch.fft.irfft2(X * (1 - m), s=x.shape[-2:], norm="ortho")
    return [lo * sd[p + "band_scale"][0], hi * sd[p + "band_scale"][1]]


def frequency_bands(sd, lr):
    return dct_bands(sd, lr) + dwt_bands(sd, lr) + fft_bands(sd, lr)


# ------------------------------------------------------------------ shared blocks
def lka_block(sd, p, x, train=False):
    """x + s1 * (n * sigmoid(BN(pw(dw21x1(dw1x21(dw5x5(n))))))), n = BN1(x); then + s2 * FFN(BN2(.))"""
    C = x.shape[1]
    n = bn_eval(sd, p + "norm1", x, train=train)
    a = conv(sd, p + "lka.local_conv", n, 2, groups=C)
    a = conv(sd, p + "lka.h_conv", a, (0, 10), groups=C)
    a = conv(sd, p + "lka.v_conv", a, (10, 0), groups=C)
    a = torch.sigmoid(bn_eval(sd, p + "lka.bn", conv(sd, p + "lka.pw_conv", a), train=train))
    x = x + sd[p + "scale1"] * (n * a)
    f = conv(sd, p + "ffn.2", F.gelu(conv(sd, p + "ffn.0", bn_eval(sd, p + "norm2", x, train=train))))
    return x + sd[p + "scale2"] * f


def mha(sd, p, x, heads):
    """nn.MultiheadAttention(batch_first) self-attention in eval mode.  x [S, T, E]."""
    S, T, E = x.shape
    hd = E // heads
    qkv = F.linear(x, sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]).reshape(S, T, 3, heads, hd)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))          # [S, heads, T, hd]
    a = ((q / math.sqrt(hd)) @ k.transpose(-2, -1)).softmax(-1)
    o = (a @ v).transpose(1, 2).reshape(S, T, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


# ------------------------------------------------------------------ phase 3
def cross_band(sd, bands, p="cross_band.", train=False):
    B, _, H, W = bands[0].shape
    T = len(bands)
    proj = torch.stack([conv(sd, p + "band_proj", b) for b in bands], 1)        # [B,T,64,H,W]
    E = proj.shape[2]
    seq = proj.permute(0, 3, 4, 1, 2).reshape(B * H * W, T, E)
    seq = seq + mha(sd, p + "band_attention.", F.layer_norm(seq, (E,), sd[p + "norm.weight"], sd[p + "norm.bias"]), 4)
    feat = seq.reshape(B, H, W, T, E).permute(0, 3, 4, 1, 2)
    return [conv(sd, p + "out_proj", lka_block(sd, p + "lka_block.", feat[:, i], train)) + bands[i] for i in range(T)]


# ------------------------------------------------------------------ phase 4
def collaborative(sd, feats, imgs, p="collaborative.", train=False):
    al = [conv(sd, f"{p}align_layers.{n}", feats[n]) for n in EXPERTS]
    h, w = min(a.shape[2] for a in al), min(a.shape[3] for a in al)
    al = [a if a.shape[2:] == (h, w) else bilinear(a, (h, w)) for a in al]
    st = torch.stack(al, 1)
    B, T, E = st.shape[:3]
    seq = st.permute(0, 3, 4, 1, 2).reshape(B * h * w, T, E)
    seq = seq + mha(sd, p + "cross_attn.", F.layer_norm(seq, (E,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]), 8)
    n2 = F.layer_norm(seq, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    seq = seq + F.linear(F.gelu(F.linear(n2, sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"])),
                         sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"])
    enh = seq.reshape(B, h, w, T, E).permute(0, 3, 4, 1, 2)
    out = []
    for i, img in enumerate(imgs):
        f = bilinear(lka_block(sd, p + "lka_global.", enh[:, i], train), img.shape[2:])
        mod = torch.sigmoid(conv(sd, f"{p}modulation.{i}.2", F.gelu(conv(sd, f"{p}modulation.{i}.0", f))))
        o = img * (1.0 + 0.2 * (mod - 0.5))
        out.append(o if train else o.clamp(0, 1))          # large_kernel_attention.py:420-423: clamp at inference only
    return out


# ------------------------------------------------------------------ phase 5
def spatial_gate(sd, p, x):
    return x * torch.sigmoid(conv(sd, p + "gate.2", F.gelu(conv(sd, p + "gate.0", x))))


def res_block(sd, p, x):
    return x + sd[p + "scale"] * conv(sd, p + "block.2", F.gelu(conv(sd, p + "block.0", x, 1)), 1)


def hierarchical(sd, imgs, p="multi_res."):
    stack = torch.cat(imgs, 1)
    Hh, Wh = stack.shape[2:]
    s1, s2 = (max(Hh // 4, 1), max(Wh // 4, 1)), (max(Hh // 2, 1), max(Wh // 2, 1))

    def stage(i, x):
        x = F.gelu(conv(sd, f"{p}stage{i}_conv.0", x, 1))
        x = F.gelu(conv(sd, f"{p}stage{i}_conv.2", x, 1))
        return res_block(sd, f"{p}stage{i}_res.", spatial_gate(sd, f"{p}stage{i}_gate.", x))

    f1 = stage(1, bilinear(stack, s1))
    f1u = bilinear(f1, s2)
    f2 = stage(2, torch.cat([f1u, bilinear(stack, s2)], 1)) + sd[p + "residual_weight_1_2"] * f1u
    f2u = bilinear(f2, (Hh, Wh))
    f3 = stage(3, torch.cat([f2u, stack], 1))
    f3 = f3 + sd[p + "residual_weight_2_3"] * f2u[:, :f3.shape[1]]
    return torch.sigmoid(conv(sd, p + "to_rgb.2", F.gelu(conv(sd, p + "to_rgb.0", f3, 1)), 1))


# ------------------------------------------------------------------ phase 6
def dynamic_selector(sd, x, p="dynamic_selector."):
    d = F.relu(conv(sd, p + "difficulty_net.0", x, 1))
    d = F.relu(conv(sd, p + "difficulty_net.2", d, 1))
    d = torch.sigmoid(conv(sd, p + "difficulty_net.4", d, 1))
    g = F.relu(conv(sd, p + "gate_net.0", x, 1))
    g = F.relu(conv(sd, p + "gate_net.2", g, 1))
    g = conv(sd, p + "gate_net.4", g)
    g = torch.sigmoid(sd[p + "temperature"] * (g - (0.7 - 0.5 * d)))
    return g / (g.sum(1, keepdim=True) + 1e-8).clamp(min=0.3), d


# ------------------------------------------------------------------ phase 7b
def edge_refine(sd, p, x):
    o = F.gelu(conv(sd, p + "conv1", x, 1))
    o = F.gelu(conv(sd, p + "conv2", o, 1))
    o = conv(sd, p + "conv3", o, 1) + conv(sd, p + "proj", x)
    return o * torch.sigmoid(conv(sd, p + "attn.attn.2", F.gelu(conv(sd, p + "attn.attn.0", o)), 1))


def laplacian_refine(sd, img, p="edge_enhance.", levels=3):
    H, W = img.shape[2:]
    pyr, cur = [], img
    for lv in range(levels):
        if lv < levels - 1:
            down = F.avg_pool2d(F.conv2d(cur, sd[p + "gaussian.kernel"], padding=2, groups=3), 2, 2)
            pyr.append(cur - bilinear(down, cur.shape[2:]))
            cur = down
        else:
            pyr.append(cur)
    lw = F.softmax(sd[p + "level_weights"], 0)
    feats = []
    for lv, lap in enumerate(pyr):
        f = edge_refine(sd, f"{p}edge_refiners.{lv}.", lap)
        if f.shape[2:] != (H, W):
            f = bilinear(f, (H, W))
        feats.append(f * lw[lv])
    edge = conv(sd, p + "fusion.2", F.gelu(conv(sd, p + "fusion.0", torch.cat(feats, 1), 1)), 1)
    gate = torch.sigmoid(conv(sd, p + "edge_gate.2", F.gelu(conv(sd, p + "edge_gate.0", torch.cat([img, edge], 1), 1)), 1))
    return (img + gate * sd[p + "edge_strength"] * edge).clamp(0, 1)


# ------------------------------------------------------------------ whole pipeline
ALL_IMPROVEMENTS = ("dynamic_expert_selection", "cross_band_attention", "adaptive_frequency_bands", "multi_resolution_fusion",
                    "collaborative_learning", "edge_enhancement")


def fusion_forward(sd, lr, imgs, feats, scale=4, return_stages=False, train=False, flags=None):
    """lr [B,3,h,w]; imgs/feats: dicts keyed drct/grl/nafnet/mamba -> final SR [B,3,4h,4w] in [0,1].
    train=True: model.train() semantics with dropout off -- BatchNorm batch statistics (running statistics in ``sd``
    updated in place), no clamp after the collaborative modulation and none on the result
    (large_kernel_attention.py:420-423, enhanced_fusion_v2.py:792-795); differentiable w.r.t. the tensors of ``sd``.
    flags: {improvement name: bool} of configs/train_config.yaml model.fusion.improvements (io.py:186-193 -> the enable_*
    arguments of CompleteEnhancedFusionSR); a disabled improvement takes the branch of _run_pipeline :696-786 that skips it."""
    on = lambda k: True if flags is None else bool(flags.get(k, True))
    B, _, h, w = lr.shape
    HR = (h * scale, w * scale)
    stages = {}
    bands = frequency_bands(sd, lr) if on("adaptive_frequency_bands") else None
    ebands, routing = bands, lr
    if on("cross_band_attention") and bands is not None:
        ebands = cross_band(sd, bands, train=train)
        routing = ebands[0] + ebands[1] + ebands[2]
    if on("collaborative_learning"):
        enh = collaborative(sd, feats, [imgs[n] for n in EXPERTS], train=train)
    else:
        enh = [imgs[n] for n in EXPERTS]
    hier = None
    if on("multi_resolution_fusion"):
        hier = hierarchical(sd, enh)
        logits = conv(sd, "freq_weight_conv.2", F.gelu(conv(sd, "freq_weight_conv.0", bilinear(routing, HR))))
        wts = logits.softmax(1)
        freq = sum(e * wts[:, i:i + 1] for i, e in enumerate(enh))
        fused = hier * 0.7 + freq * 0.3
    else:
        fused = conv(sd, "simple_fusion", torch.cat(enh, 1))
    if on("dynamic_expert_selection"):
        gates, diff = dynamic_selector(sd, routing)
        g_hr = bilinear(gates, HR)
        dyn = sum(e * g_hr[:, i:i + 1] for i, e in enumerate(enh)) / (g_hr.sum(1, keepdim=True) + 1e-8)
        bw = 0.3 + 0.4 * bilinear(diff, HR)
        fused = (1 - bw) * fused + bw * dyn
    r = fused
    for i in range(0, 10, 2):
        r = F.gelu(conv(sd, f"refine.{i}", r, 1))
    fused = fused + 0.1 * conv(sd, "refine.10", r, 1)
    edged = laplacian_refine(sd, fused) if on("edge_enhancement") else fused
    out = edged + sd["residual_scale"] * bilinear(lr, HR)
    if not train:
        out = out.clamp(0, 1)
    if return_stages:
        stages.update(bands=bands, ebands=ebands, routing=routing, enh=enh, hier=hier, fused_pre_refine=None,
                      refined=fused, edged=edged)
        return out, stages
    return out
