"""7-phase frequency-guided fusion (CompleteEnhancedFusionSR, eval mode) on the HIP kernels, host side.

Mirrors src/models/enhanced_fusion_v2.py (forward_with_precomputed :642, _run_pipeline :681), with the sub-modules
of multi_domain_frequency.py, large_kernel_attention.py, hierarchical_fusion.py and edge_enhancement.py.
Same state_dict keys as the reference.  GPU-first re-organisation (results unchanged up to fp32 rounding):

* the 9 frequency bands live in ONE token tensor bands[pixel][band][4] -- band_proj / the cross-band MHA run as
  token GEMMs over it without any stack/permute copies; only bands 0-2 go through the LKA block because
  `routing_lr = e0 + e1 + e2` is the only consumer (enhanced_fusion_v2.py:713);
* eval-mode BatchNorms are folded into the adjacent 1x1 convolutions where that is exact;
* concatenations are channel slices of pre-allocated buffers (producers write in place);
* phase 4's 1x1 conv (128->32) is applied before the bilinear upsample (both linear), the HR-resolution
  modulation / routing / gating / final residual are fused elementwise kernels.

The six improvement switches of configs/train_config.yaml (model.fusion.improvements -> the enable_* arguments,
io.py:186-193) select the same alternative branches as _run_pipeline :696-786: a disabled improvement's sub-module is
neither expected in the state_dict nor run.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .common import dev, tokens
from .ops import ACT_GELU, ACT_RELU, ACT_SIGMOID
from .weights import improvement_flags

EXPERTS = ("drct", "grl", "nafnet", "mamba")


def _bn_affine(sd, p, eps=1e-5):
    s = sd[p + ".weight"].float() / torch.sqrt(sd[p + ".running_var"].float() + eps)
    return s, sd[p + ".bias"].float() - sd[p + ".running_mean"].float() * s


class _LKABlock:
    """LKABlock (large_kernel_attention.py:112-149) with BN folding."""

    def __init__(self, sd, p, device):
        self.s1, self.s2 = float(sd[p + "scale1"]), float(sd[p + "scale2"])
        a, b = _bn_affine(sd, p + "norm1")
        self.n1 = (dev(a, device), dev(b, device))
        self.dw5 = ops.pack_dwconv(sd[p + "lka.local_conv.weight"], None, device)
        self.dwh = ops.pack_dwconv(sd[p + "lka.h_conv.weight"], None, device)
        self.dwv = ops.pack_dwconv(sd[p + "lka.v_conv.weight"], None, device)
        a, b = _bn_affine(sd, p + "lka.bn")
        self.pw = ops.pack_conv(sd[p + "lka.pw_conv.weight"].float() * a[:, None, None, None], b, device)
        a, b = _bn_affine(sd, p + "norm2")
        w0 = sd[p + "ffn.0.weight"].float()
        self.f0 = ops.pack_conv(w0 * a[None, :, None, None], sd[p + "ffn.0.bias"].float() + w0[:, :, 0, 0] @ b, device)
        self.f2 = ops.pack_conv(sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"], device)

    def __call__(self, x):
        n = ops.unary(x, cscale=self.n1[0], cbias=self.n1[1])
        a = ops.dwconv2d(ops.dwconv2d(ops.dwconv2d(n, self.dw5), self.dwh), self.dwv)
        a = ops.conv2d(a, self.pw, act=ACT_SIGMOID)
        x1 = ops.mul_add(n, a, c=x, alpha=self.s1)                       # x + s1 * n * sigmoid(...)
        h = ops.conv2d(x1, self.f0, act=ACT_GELU)
        return ops.conv2d(h, self.f2, res=x1, cscale=self.s2)


class _MHA:
    """nn.MultiheadAttention (eval) as in_proj GEMM -> per-pixel attention kernel -> out_proj GEMM (+ residual)."""

    def __init__(self, sd, p, device, heads, ln=None):
        """ln = (gamma, beta) of the LayerNorm in front of the attention: folded into in_proj's token GEMM (one launch)"""
        self.heads = heads
        self.inp = ops.pack_conv(sd[p + "in_proj_weight"], sd[p + "in_proj_bias"], device)
        E = sd[p + "in_proj_weight"].shape[1]
        self.inp_t = ops.pack_tok_gemm(sd[p + "in_proj_weight"], sd[p + "in_proj_bias"], device, ln=ln) \
            if ln is not None and ops.tok_gemm_ok(E, 3 * E) else None
        self.ln_fused = self.inp_t is not None
        self.out = ops.pack_conv(sd[p + "out_proj.weight"], sd[p + "out_proj.bias"], device)
        self.E = self.out.N

    def __call__(self, normed, resid, S, T):
        """normed: the LayerNorm-ed tokens -- or, with `ln_fused`, the raw tokens (the norm rides in the in_proj kernel)"""
        if self.inp_t is not None and self.ln_fused and ops.tok_enabled():
            qkv = ops.tok_gemm(normed, self.inp_t)
        else:
            qkv = ops.linear(normed, self.inp)
        a = ops.pixel_mha(qkv, S, T, self.E, self.heads)
        return ops.linear(a, self.out, res=resid)


class FusionNet:
    def __init__(self, sd, device, scale=4, flags=None):
        self.device, self.scale = device, scale
        self.flags = improvement_flags(flags)
        on = self.flags.get
        g = lambda k: sd[k].float()
        self.refine = [ops.pack_conv(sd[f"refine.{i}.weight"], sd[f"refine.{i}.bias"], device, cin_pad=4 if i == 0 else None)
                       for i in range(0, 12, 2)]
        self.residual_scale = dev(g("residual_scale").reshape(1), device)
        self._fft_cache = {}
        if on("adaptive_frequency_bands"):
            self._init_bands(sd, device, g)
        # without the bands phase 3 has nothing to enhance: routing = the LR image (enhanced_fusion_v2.py:706-718)
        self.use_cross_band = on("cross_band_attention") and on("adaptive_frequency_bands")
        if self.use_cross_band:
            self._init_cross_band(sd, device, g)
        if on("collaborative_learning"):
            self._init_collaborative(sd, device, g)
        if on("multi_resolution_fusion"):
            self._init_multi_res(sd, device, g)
        else:
            self.simple = ops.pack_conv(sd["simple_fusion.weight"], sd["simple_fusion.bias"], device)
        if on("dynamic_expert_selection"):
            self._init_selector(sd, device, g)
        if on("edge_enhancement"):
            self._init_edge(sd, device, g)

    def _init_bands(self, sd, device, g):
        # ---- phase 2
        self.dct_D = dev(g("freq_decomp.dct.dct_basis"), device)
        self.dct_masks = dev(torch.stack([g("freq_decomp.dct.low_mask"), g("freq_decomp.dct.mid_mask"),
                                          g("freq_decomp.dct.high_mask")]).reshape(3, 64), device)
        self.dct_scale = dev(g("freq_decomp.dct.band_scale"), device)
        self.dwt_lo = dev(g("freq_decomp.dwt.lo_row")[0].reshape(-1), device)
        self.dwt_hi = dev(g("freq_decomp.dwt.hi_row")[0].reshape(-1), device)
        self.dwt_scale = [float(v) for v in g("freq_decomp.dwt.subband_scale")]
        logits = torch.zeros(1, 64, 64, 4)
        logits[..., 0] = g("freq_decomp.fft.freq_mask_logits")[0, 0]
        self.fft_logits = logits.to(device)[..., :1]
        self.fft_temp = max(float(g("freq_decomp.fft.temperature")), 1.0)
        self.fft_scale = dev(g("freq_decomp.fft.band_scale"), device)

    def _init_cross_band(self, sd, device, g):
        # ---- phase 3
        p = "cross_band."
        self.band_proj = ops.pack_conv(sd[p + "band_proj.weight"], sd[p + "band_proj.bias"], device, cin_pad=4)
        self.cb_norm = (dev(g(p + "norm.weight"), device), dev(g(p + "norm.bias"), device))
        self.cb_mha = _MHA(sd, p + "band_attention.", device, 4, ln=(sd[p + "norm.weight"], sd[p + "norm.bias"]))
        self.cb_lka = _LKABlock(sd, p + "lka_block.", device)
        self.cb_out = ops.pack_conv(sd[p + "out_proj.weight"], sd[p + "out_proj.bias"], device)

    def _init_collaborative(self, sd, device, g):
        # ---- phase 4
        p = "collaborative."
        self.align = {n: ops.pack_conv(sd[f"{p}align_layers.{n}.weight"], sd[f"{p}align_layers.{n}.bias"], device)
                      for n in EXPERTS}
        self.co_mha = _MHA(sd, p + "cross_attn.", device, 8, ln=(sd[p + "norm1.weight"], sd[p + "norm1.bias"]))
        # norm2 + ffn + residual as one kernel (the Swin Mlp form: x + fc2(GELU(fc1(LayerNorm(x)))))
        self.co_ffn = ops.pack_tok_chain(sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"], sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"],
                                         device, mode=0, ln=(sd[p + "norm2.weight"], sd[p + "norm2.bias"])) \
            if ops.tok_chain_ok(128, 128, 0) else None
        self.co_n1 = (dev(g(p + "norm1.weight"), device), dev(g(p + "norm1.bias"), device))
        self.co_n2 = (dev(g(p + "norm2.weight"), device), dev(g(p + "norm2.bias"), device))
        self.co_f0 = ops.pack_conv(sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"], device)
        self.co_f2 = ops.pack_conv(sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"], device)
        self.co_lka = _LKABlock(sd, p + "lka_global.", device)
        self.mod0 = [ops.pack_conv(sd[f"{p}modulation.{i}.0.weight"], sd[f"{p}modulation.{i}.0.bias"], device) for i in range(4)]
        self.mod2 = [(dev(g(f"{p}modulation.{i}.2.weight").reshape(3, 32), device), dev(g(f"{p}modulation.{i}.2.bias"), device))
                     for i in range(4)]

    def _init_multi_res(self, sd, device, g):
        # ---- phase 5
        p = "multi_res."
        pc = lambda k, **kw: ops.pack_conv(sd[p + k + ".weight"], sd.get(p + k + ".bias"), device, **kw)
        self.stage = []
        for s in (1, 2, 3):
            self.stage.append(dict(c0=pc(f"stage{s}_conv.0"), c2=pc(f"stage{s}_conv.2"), g0=pc(f"stage{s}_gate.gate.0"),
                                   g2=pc(f"stage{s}_gate.gate.2"), r0=pc(f"stage{s}_res.block.0"),
                                   r2=pc(f"stage{s}_res.block.2"), rs=float(g(p + f"stage{s}_res.scale"))))
        self.rw12, self.rw23 = float(g(p + "residual_weight_1_2")), float(g(p + "residual_weight_2_3"))
        self.rgb0, self.rgb2 = pc("to_rgb.0"), pc("to_rgb.2")
        self.fw = dev(torch.cat([g("freq_weight_conv.0.weight").reshape(-1), g("freq_weight_conv.0.bias"),
                                 g("freq_weight_conv.2.weight").reshape(-1), g("freq_weight_conv.2.bias")]), device)

    def _init_selector(self, sd, device, g):
        # ---- phase 6
        p = "dynamic_selector."
        pc = lambda k, **kw: ops.pack_conv(sd[p + k + ".weight"], sd.get(p + k + ".bias"), device, **kw)
        self.dn = [pc("difficulty_net.0", cin_pad=4), pc("difficulty_net.2"), pc("difficulty_net.4")]
        self.gn = [pc("gate_net.0", cin_pad=4), pc("gate_net.2"), pc("gate_net.4")]
        self.temperature = dev(g(p + "temperature").reshape(1), device)

    def _init_edge(self, sd, device, g):
        # ---- phase 7b
        p = "edge_enhance."
        pc = lambda k, **kw: ops.pack_conv(sd[p + k + ".weight"], sd.get(p + k + ".bias"), device, **kw)
        gk = torch.zeros(4, 1, 5, 5)
        gk[:3] = g(p + "gaussian.kernel")
        self.gauss = ops.pack_dwconv(gk, None, device)
        self.level_w = [float(v) for v in torch.softmax(g(p + "level_weights"), 0)]
        self.refiners = []
        for i in range(3):
            q = f"edge_refiners.{i}."
            self.refiners.append(dict(c1=pc(q + "conv1", cin_pad=4), c2=pc(q + "conv2"), c3=pc(q + "conv3"),
                                      pj=pc(q + "proj", cin_pad=4), a0=pc(q + "attn.attn.0"), a2=pc(q + "attn.attn.2")))
        self.ef0, self.ef2 = pc("fusion.0"), pc("fusion.2")
        self.eg0, self.eg2 = pc("edge_gate.0", cin_pad=8), pc("edge_gate.2")
        self.edge_strength = dev(g(p + "edge_strength").reshape(1), device)

    # ------------------------------------------------------------------------------------------ phase 2
    def prepare(self, h, w):
        """Build (once per image size) the cached input-independent tables on the CURRENT stream and make every later user
        wait for them: the cache entry carries an event that `_fft_tables` makes the consuming stream wait on, so a hit
        from another stream lane (or from a side stream) can never read a table that is still being written."""
        if self.flags["adaptive_frequency_bands"]:
            self._fft_tables(h, w)

    def _fft_tables(self, h, w):
        key = (h, w)
        ent = self._fft_cache.get(key)
        if ent is not None:
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(self.device).wait_event(ent[3])
            return ent[:3]

        def tw(n):
            j = torch.arange(n, dtype=torch.float64) * (2 * math.pi / n)
            return torch.stack([torch.cos(j), torch.sin(j)], 1).float().contiguous().to(self.device)
        wf = w // 2 + 1
        mask = torch.empty(1, h, wf, 1, device=self.device)
        ops.bilinear(self.fft_logits, h, wf, out=mask)
        ops.unary(mask, act=ACT_SIGMOID, pre=self.fft_temp, out=mask)
        tabs = (tw(w), tw(h), mask)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._fft_cache[key] = tabs + (done,)
        return tabs

    def frequency_bands(self, lr):
        """lr [B,h,w,3] -> bands [B,h,w,36] = 9 bands x (3 channels + zero pad)."""
        B, h, w, _ = lr.shape
        bands = ops.zeros(B, h, w, 36, device=lr.device)
        ops.dct_bands(lr, self.dct_D, self.dct_masks, self.dct_scale, bands)
        sub = ops.dwt_db4(lr, self.dwt_lo, self.dwt_hi)
        for i in range(4):
            ops.bilinear(sub[..., 4 * i:4 * i + 4], h, w, mul=self.dwt_scale[i], out=bands[..., 12 + 4 * i:16 + 4 * i])
        twW, twH, mask = self._fft_tables(h, w)
        ops.fft_bands(lr, twW, twH, mask, self.fft_scale, bands)
        return bands

    # ------------------------------------------------------------------------------------------ phase 3
    def cross_band_routing(self, bands):
        """-> routing_lr [B,h,w,3] = sum of the first three enhanced bands."""
        B, h, w, _ = bands.shape
        P = B * h * w
        rows = bands.reshape(P * 9, 4)
        proj = ops.linear(rows, self.band_proj)                                     # [P*9, 64]
        fused = self.cb_mha.ln_fused and ops.tok_enabled()
        attn = self.cb_mha(proj if fused else ops.layernorm(proj, *self.cb_norm), proj, P, 9)          # + residual
        routing = None
        for i in range(3):
            feat = attn.as_strided((B, h, w, 64), (h * w * 576, w * 576, 576, 1), attn.storage_offset() + 64 * i)
            e = ops.conv2d(self.cb_lka(feat), self.cb_out, res=bands[..., 4 * i:4 * i + 3])
            routing = e if routing is None else ops.scale_add(routing, e)
        return routing

    # ------------------------------------------------------------------------------------------ phase 4
    def collaborative(self, feats, imgs, enh):
        """feats: dict of [B,h,w,C_e]; imgs: dict of [B,Hh,Wh,3]; writes the 4 modulated images into enh [B,Hh,Wh,12]."""
        B, h, w, _ = feats["drct"].shape
        P = B * h * w
        st = torch.empty(P * 4, 128, device=self.device)
        for e, n in enumerate(EXPERTS):
            view = st.as_strided((B, h, w, 128), (h * w * 512, w * 512, 512, 1), 128 * e)
            ops.conv2d(feats[n], self.align[n], out=view)
        fused = self.co_mha.ln_fused and ops.tok_enabled()
        s1 = self.co_mha(st if fused else ops.layernorm(st, *self.co_n1), st, P, 4)
        if self.co_ffn is not None and ops.tok_enabled():
            s2 = ops.tok_chain(s1, self.co_ffn, res=s1)
        else:
            hdn = ops.linear(ops.layernorm(s1, *self.co_n2), self.co_f0, act=ACT_GELU)
            s2 = ops.linear(hdn, self.co_f2, res=s1)
        for e, n in enumerate(EXPERTS):
            view = s2.as_strided((B, h, w, 128), (h * w * 512, w * 512, 512, 1), s2.storage_offset() + 128 * e)
            t_lr = ops.conv2d(self.co_lka(view), self.mod0[e])                      # 1x1 128->32 at LR
            ops.modulate(t_lr, self.mod2[e][0], self.mod2[e][1], imgs[n], enh[..., 3 * e:3 * e + 3])

    # ------------------------------------------------------------------------------------------ phase 5
    def _stage(self, i, x):
        s = self.stage[i]
        x = ops.conv2d(ops.conv2d(x, s["c0"], act=ACT_GELU), s["c2"], act=ACT_GELU)
        gate = ops.conv2d(ops.conv2d(x, s["g0"], act=ACT_GELU), s["g2"], act=ACT_SIGMOID)
        x = ops.mul_add(x, gate, row_broadcast=True)
        return ops.conv2d(ops.conv2d(x, s["r0"], act=ACT_GELU), s["r2"], res=x, cscale=s["rs"])

    def hierarchical(self, cat3):
        """cat3 [B,Hh,Wh,76]: channels 64..75 hold the expert stack; channels 0..63 are filled here."""
        B, Hh, Wh, _ = cat3.shape
        enh = cat3[..., 64:76]
        s1, s2 = (max(Hh // 4, 1), max(Wh // 4, 1)), (max(Hh // 2, 1), max(Wh // 2, 1))
        f1 = self._stage(0, ops.bilinear(enh, *s1))
        cat2 = torch.empty(B, s2[0], s2[1], 76, device=self.device)
        f1u = ops.bilinear(f1, *s2, out=cat2[..., :64])
        ops.bilinear(enh, *s2, out=cat2[..., 64:76])
        f2 = self._stage(1, cat2)
        f2 = ops.scale_add(f2, f1u, beta=self.rw12)
        f2u = ops.bilinear(f2, Hh, Wh, out=cat3[..., :64])
        f3 = self._stage(2, cat3)
        f3 = ops.scale_add(f3, f2u[..., :32], beta=self.rw23)
        return ops.conv2d(ops.conv2d(f3, self.rgb0, act=ACT_GELU), self.rgb2, act=ACT_SIGMOID)

    # ------------------------------------------------------------------------------------------ phase 6
    def selector(self, routing):
        r4 = ops.widen(routing, 4)
        d = ops.conv2d(ops.conv2d(ops.conv2d(r4, self.dn[0], act=ACT_RELU), self.dn[1], act=ACT_RELU), self.dn[2],
                       act=ACT_SIGMOID)
        raw = ops.conv2d(ops.conv2d(ops.conv2d(r4, self.gn[0], act=ACT_RELU), self.gn[1], act=ACT_RELU), self.gn[2])
        return ops.selector_gates(raw, d, self.temperature), d

    # ------------------------------------------------------------------------------------------ phase 7b
    def _edge_refine(self, lv, x, out=None):
        r = self.refiners[lv]
        x4 = ops.widen(x, 4)
        o = ops.conv2d(ops.conv2d(x4, r["c1"], act=ACT_GELU), r["c2"], act=ACT_GELU)
        o = ops.conv2d(o, r["c3"], res=ops.conv2d(x4, r["pj"]))
        att = ops.conv2d(ops.conv2d(o, r["a0"], act=ACT_GELU), r["a2"], act=ACT_SIGMOID)
        return ops.mul_add(o, att, row_broadcast=True, alpha=self.level_w[lv], out=out)   # level weight folded in

    def laplacian_refine(self, sr, lr, out):
        """sr [B,Hh,Wh,3] (ld 4, pad 0) -> out = clamp(clamp(sr + gate*strength*edge) + residual_scale*bilinear(lr))."""
        B, Hh, Wh, _ = sr.shape
        pyr, cur = [], sr
        for lv in range(3):
            if lv < 2:
                down = ops.avgpool2(ops.dwconv2d(ops.widen(cur, 4), self.gauss))
                up = ops.bilinear(down, cur.shape[1], cur.shape[2])
                pyr.append(ops.scale_add(ops.widen(cur, 4), up, beta=-1.0)[..., :3])
                cur = down[..., :3]
            else:
                pyr.append(cur)
        feats = torch.empty(B, Hh, Wh, 96, device=self.device)
        for lv, lap in enumerate(pyr):
            if lv == 0:
                self._edge_refine(0, lap, out=feats[..., :32])
            else:
                ops.bilinear(self._edge_refine(lv, lap), Hh, Wh, out=feats[..., 32 * lv:32 * lv + 32])
        cat6 = ops.zeros(B, Hh, Wh, 8, device=self.device)
        ops.unary(sr, out=cat6[..., :3])
        edge = ops.conv2d(ops.conv2d(feats, self.ef0, act=ACT_GELU), self.ef2, out=cat6[..., 3:6])
        gate = ops.conv2d(ops.conv2d(cat6, self.eg0, act=ACT_GELU), self.eg2, act=ACT_SIGMOID)
        ops.edge_final(sr, edge, gate, self.edge_strength, lr, self.residual_scale, out)
        return out

    # ------------------------------------------------------------------------------------------ whole pipeline
    def lr_phases(self, lr):
        """The part of the pipeline that needs only the LR image: phase 2 (bands), phase 3 (cross-band routing) and the
        selector's gates of phase 6.  The engine runs it on a side stream while the four experts compute."""
        bands = self.frequency_bands(lr) if self.flags["adaptive_frequency_bands"] else None
        routing = self.cross_band_routing(bands) if self.use_cross_band else lr
        gates, diff = self.selector(routing) if self.flags["dynamic_expert_selection"] else (None, None)
        return bands, routing, gates, diff

    def __call__(self, lr, imgs, feats, return_stages=False, pre=None):
        """lr [B,h,w,3]; imgs: 4 x [B,4h,4w,3]; feats: [B,h,w,180|180|64|180] -> SR [B,4h,4w,3] in [0,1].
        pre: the result of lr_phases(lr) if the caller already computed it."""
        B, h, w, _ = lr.shape
        Hh, Wh = h * self.scale, w * self.scale
        bands, routing, gates, diff = pre if pre is not None else self.lr_phases(lr)
        cat3 = torch.empty(B, Hh, Wh, 76, device=self.device)
        enh = cat3[..., 64:76]
        if self.flags["collaborative_learning"]:
            self.collaborative(feats, imgs, enh)
        else:                                                   # the expert images themselves are fused (:721-726)
            for e, n in enumerate(EXPERTS):
                ops.unary(imgs[n], out=enh[..., 3 * e:3 * e + 3])
        fused = torch.empty(B, Hh, Wh, 4, device=self.device)
        if self.flags["multi_resolution_fusion"]:
            hier = self.hierarchical(cat3)
            ops.fusion_route(enh, hier, routing, self.fw, gates, diff, fused)
        else:                                                   # simple_fusion: 1x1 conv over the 12 stacked channels (:748-750)
            hier = ops.conv2d(enh, self.simple)
            ops.fusion_route(enh, hier, routing, None, gates, diff, fused)
        if ops.PLANES_AUTO and ops.GEMM_MODE == "bf16x3":   # the whole 6-conv stack runs on bf16 hi/lo planes
            r = ops.split_planes(fused[..., :3])
            thin = ops.thin3_ok(self.refine[-1], B * Hh * Wh)        # the 128 -> 3 head reads an fp32 map
            for i, cv in enumerate(self.refine[:-1]):
                last = thin and i == len(self.refine) - 2
                r = ops.conv2d(r, cv, act=ACT_GELU, out_planes=None if last else True, want_f32=last)
        else:
            r = fused
            for cv in self.refine[:-1]:
                r = ops.conv2d(r, cv, act=ACT_GELU)
        refined = ops.conv2d(r, self.refine[-1], res=fused[..., :3], cscale=0.1)
        out = ops.new_map(B, Hh, Wh, 3, self.device)
        if self.flags["edge_enhancement"]:
            self.laplacian_refine(refined, lr, out)
        else:
            ops.edge_final(refined, None, None, None, lr, self.residual_scale, out)
        if return_stages:
            return out, dict(bands=bands, routing=routing, enh=enh, hier=hier, fused=fused, refined=refined)
        return out
