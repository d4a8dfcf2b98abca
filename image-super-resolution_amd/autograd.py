"""A tape of HIP launches: the host-side stand-in for torch.autograd on the fusion network's training step.

The reference trains ``CompleteEnhancedFusionSR`` with ``loss.backward()`` (train.py:323-336): ATen records a graph of the
forward operators and replays their derivative formulas.  Here every differentiable operator of the fusion net is a method
of ``Tape``: it launches the forward kernel(s) through the C ABI right away and appends a closure that launches the matching
backward kernel(s); ``Tape.backward()`` runs the closures in reverse.  torch supplies device memory and views only -- no
arithmetic, no torch.autograd (the product path runs under ``torch.no_grad``).

Values are the channels-last maps of ops.py ([B,H,W,C] or [M,C] matrices, possibly channel-slice views of a wider buffer).
Parameters live in ONE flat fp32 buffer with a flat gradient buffer beside it (train.FusionOptimizer); parameter gradients
are accumulated in place by the kernels (``dst +=``), so a module applied several times per step (the shared LKA block)
or several micro-batches (accumulation_steps, train.py:332) need nothing extra.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

from . import hip, ops
from .ops import ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, _mat, _ptr, _stream

ACT_CLAMP01, ACT_CLAMP_MIN = 6, 7        # codes of ffsr_act_bwd_f32 only
N_PARTIAL = 1024
WGRAD_BF16X3 = os.environ.get("FFSR_WGRAD_BF16X3", "1") != "0"   # =0: every weight gradient on the exact f32 MFMA kernels (A/B runs)


class Var:
    """A value on the tape: ``v`` the tensor, ``g`` its gradient (same logical shape; None until something flowed back),
    ``req`` False for constants (cached expert images / features, the LR image)."""
    __slots__ = ("v", "g", "req", "gown")

    def __init__(self, v: torch.Tensor, req: bool = True):
        self.v, self.g, self.req, self.gown = v, None, req, False

    @property
    def shape(self):
        return tuple(self.v.shape)


class Param:
    """A learnable tensor: ``v`` / ``g`` are views into the flat parameter / gradient buffers (nn.Parameter layout)."""
    __slots__ = ("v", "g", "name")

    def __init__(self, name, v, g):
        self.name, self.v, self.g = name, v, g


class ConvP:
    """nn.Conv2d / nn.Linear parameters + the per-step packed operands (forward and input-gradient operators)."""

    def __init__(self, weight: Param, bias: Optional[Param], device, pad=None, cin_pad=None):
        w = weight.v
        self.weight, self.bias = weight, bias
        self.N, self.Cin = w.shape[0], w.shape[1]
        self.KH, self.KW = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
        self.pad = self.KH // 2 if pad is None else pad
        T = self.KH * self.KW
        self.fwd = _alloc_conv(self.N, self.Cin, cin_pad or ops.pad4(self.Cin), self.KH, self.KW, self.pad, device,
                               None if bias is None else bias.v)
        self.bwd = _alloc_conv(self.Cin, self.N, ops.pad4(self.N), self.KH, self.KW, self.KH - 1 - self.pad, device, None)

    def repack(self):
        """weights changed (optimiser step): rebuild both packed operands, one launch each"""
        w = self.weight.v
        for cv, tr in ((self.fwd, 0), (self.bwd, 1)):
            hip.call("ffsr_pack_conv_f32", _ptr(w), self.N, self.Cin, self.KH, self.KW, tr, _ptr(cv.wgt), cv.Cin,
                     _ptr(cv.whi), _ptr(cv.wlo), cv.whi.shape[0], cv.whi.shape[1], _stream())


def _alloc_conv(N, Cin_true, Cp, KH, KW, pad, device, bias) -> ops.Conv:
    K = KH * KW * Cp
    # with Cin a multiple of 32 the split-bf16 planes [rows, taps * Cp] are also the planes kernels' weight layout (every tap
    # padded to 32 channels): rows padded to x768 (any column tile) and the same buffers serve ffsr_conv2d_planes
    planes_ok = Cin_true % 32 == 0 and Cp == Cin_true and KH * KW > 1
    Np, Kp = (N + 767) // 768 * 768 if planes_ok else (N + 127) // 128 * 128, (K + 31) // 32 * 32
    hi = torch.empty(Np, Kp, dtype=torch.bfloat16, device=device)
    lo = torch.empty(Np, Kp, dtype=torch.bfloat16, device=device)
    return ops.Conv(torch.empty(N, K, device=device), bias, N, Cp, KH, KW, 1, pad, Cin_true, hi, lo,
                    hi if planes_ok else None, lo if planes_ok else None, Cp if planes_ok else 0)


class DwP:
    """depthwise nn.Conv2d(groups=C, bias=False) weight, or a fixed kernel (weight given as a tensor: no gradient)"""

    def __init__(self, weight, device, pad):
        self.param = weight if isinstance(weight, Param) else None
        w = weight.v if self.param is not None else weight
        self.C, _, self.KH, self.KW = w.shape
        self.w = w
        self.ph, self.pw = pad
        T = self.KH * self.KW
        self.fwd = ops.DwConv(torch.empty(T, self.C, device=device), None, self.KH, self.KW, self.ph, self.pw)
        self.bwd = ops.DwConv(torch.empty(T, self.C, device=device), None, self.KH, self.KW, self.KH - 1 - self.ph,
                              self.KW - 1 - self.pw)
        self.repack()

    def repack(self):
        for dw, flip in ((self.fwd, 0), (self.bwd, 1)):
            hip.call("ffsr_pack_dwconv_f32", _ptr(self.w), self.C, self.KH, self.KW, flip, _ptr(dw.w), _stream())


class BnP:
    def __init__(self, gamma: Param, beta: Param, run_mean: torch.Tensor, run_var: torch.Tensor):
        self.gamma, self.beta, self.run_mean, self.run_var = gamma, beta, run_mean, run_var
        self.calls = 0                      # num_batches_tracked


def _wgrad_scratch(p) -> int:
    """floats of scratch for a layer's weight gradient: room for 256 pixel splits of the MFMA kernels, and for the 1024 block
    partials of the thin-layer kernels (N <= 4 or Cin <= 4: small tiles -- with 256 they ran on a quarter of the chip)"""
    per = p.weight.v.numel() + p.N
    return max(per * 256, min(per * 1024, 1 << 23), 1 << 20)


def conv_tile(M: int, N: int, taps: int, cin: int = 0) -> int:
    """explicit tile_hint of ops.conv2d for fp32-map inputs: the split-bf16 kernel where the default mode would take it,
    else the exact f32 kernel (also FFSR_GEMM_MODE=f32).  0 = ops.conv2d's own choice: for the wide k x k layers at HR (the refine
    stack, forward and input gradient) that is one split pass + the planes tap-strip kernel (2.2 vs 2.8 ms per layer at config 5)."""
    if ops.GEMM_MODE == "bf16x3" and ops.PLANES_AUTO and taps > 1 and cin >= 32 and cin % 32 == 0 and N > 64 and M >= 16384:
        return 0
    if ops.GEMM_MODE == "bf16x3" and M > 64 * 24 and taps <= 32:
        return 32 if N <= 32 else 64
    if N <= 32:
        return 13
    return 4 if M <= 64 * 24 else 12


class Tape:
    def __init__(self, device):
        self.device = torch.device(device)
        self._bw = []
        self.enabled = True
        self.seed, self.step, self._draws = 0, 0, 0     # dropout draws: (seed, step, index of the draw within the step)

    # ------------------------------------------------------------------------------------------ plumbing
    def _rec(self, fn):
        if self.enabled:
            self._bw.append(fn)

    def backward(self):
        for fn in reversed(self._bw):
            fn()
        self._bw.clear()

    def clear(self):
        self._bw.clear()

    def new_step(self, step: int):
        """start a forward pass: dropout masks of this pass are drawn from (seed, step, 0..)"""
        self.step, self._draws = int(step), 0

    class _NoGrad:
        def __init__(self, tape):
            self.tape = tape

        def __enter__(self):
            self.prev, self.tape.enabled = self.tape.enabled, False

        def __exit__(self, *a):
            self.tape.enabled = self.prev

    def no_grad(self):
        """forward only (the BatchNorm statistics of the bands whose output nothing consumes)"""
        return Tape._NoGrad(self)

    def acc(self, var: Var, g: torch.Tensor, owned=True):
        """var.g += g.  `owned`: g is a fresh buffer (or a region nobody else reads) that later accumulation may overwrite."""
        if not var.req:
            return
        if var.g is None:
            var.g, var.gown = g, owned
        elif var.gown:
            ops.scale_add(var.g, g, out=var.g)
        else:
            var.g, var.gown = ops.scale_add(var.g, g), True

    def _partial(self, n=N_PARTIAL):
        return torch.empty(n, dtype=torch.float64, device=self.device)

    def dot_into(self, dst: torch.Tensor, a, b=None, scale=1.0):
        """dst[0] += scale * sum(a * b): gradient of a learnable scalar (dst: 0-dim / 1-element view of the flat gradient)"""
        _, M, C, lda = _mat(a)
        hip.call("ffsr_dot_acc_f32", _ptr(a), lda, _ptr(b), 0 if b is None else _mat(b)[3], M, C, _ptr(self._partial()),
                 N_PARTIAL, _ptr(dst), float(scale), 1, _stream())

    def coldot_into(self, dst: torch.Tensor, a, b=None, ostride=1, scale=1.0):
        """dst[c * ostride] += scale * sum_m a[m, c] * b[m, c]"""
        _, M, C, lda = _mat(a)
        nchunk = max(1, min(512, M // 256))
        part = torch.empty(nchunk * C, device=self.device)
        hip.call("ffsr_coldot_acc_f32", _ptr(a), lda, _ptr(b), 0 if b is None else _mat(b)[3], M, C, _ptr(part), nchunk,
                 _ptr(dst), ostride, float(scale), 1, _stream())

    def act_bwd(self, dy, ref, act, slope=0.0, from_output=False, alpha=1.0):
        _, M, C, ldy = _mat(dy)
        dx = ops._like(dy)
        hip.call("ffsr_act_bwd_f32", _ptr(dy), ldy, _ptr(ref), _mat(ref)[3], _ptr(dx), _mat(dx)[3], M, C, act, float(slope),
                 int(from_output), float(alpha), 0, _stream())
        return dx

    # ------------------------------------------------------------------------------------------ views
    def join(self, parts: Sequence[Var], whole: torch.Tensor, selectors) -> Var:
        """`whole` already holds the parts (part i was produced with out=selectors[i](whole)): the concatenations of the
        reference (torch.cat / torch.stack) without a copy.  backward: part.g += the same window of whole.g."""
        y = Var(whole)

        def bw():
            if y.g is None:
                return
            for p, sel in zip(parts, selectors):
                self.acc(p, sel(y.g), owned=True)
        self._rec(bw)
        return y

    def split(self, x: Var, selectors) -> List[Var]:
        """windows (channel slices / token-interleaved rows) of x.v as separate Vars.  backward: x.g = zeros, then every
        part's gradient is added into its window."""
        parts = [Var(sel(x.v), x.req) for sel in selectors]

        def bw():
            if not x.req or all(p.g is None for p in parts):
                return
            g = ops.zeros(*x.v.shape, device=self.device)
            for p, sel in zip(parts, selectors):
                if p.g is not None:
                    gv = sel(g)
                    ops.scale_add(gv, p.g, out=gv)
            self.acc(x, g)
        self._rec(bw)
        return parts

    # ------------------------------------------------------------------------------------------ dense conv / linear
    def conv(self, x: Var, p: ConvP, act=ACT_NONE, slope=0.0, out=None) -> Var:
        """nn.Conv2d (stride 1, "same") / nn.Linear + optional activation.  GELU keeps the pre-activation for its backward;
        ReLU / LeakyReLU / sigmoid differentiate from the output."""
        xv = x.v
        is2d = xv.dim() == 2
        xm = ops.as_map(xv) if is2d else xv
        B, H, W, _ = xm.shape
        M = B * H * W
        hint = 0 if ops.thin3_ok(p.fwd, M, xm) else conv_tile(M, p.N, p.KH * p.KW, p.Cin)   # (0: the N <= 4 heads take the streaming kernel)
        fused = act if act in (ACT_RELU, ACT_LRELU, ACT_SIGMOID) else ACT_NONE
        om = None if out is None else (ops.as_map(out) if out.dim() == 2 else out)
        if act == ACT_GELU:
            z = ops.conv2d(xm, p.fwd, tile_hint=hint)
            ym = ops.unary(z, act=ACT_GELU, out=om)
            saved = z
        else:
            ym = ops.conv2d(xm, p.fwd, act=fused, slope=slope, out=om, tile_hint=hint)
            saved = ym
        y = Var(_unmap(ym) if is2d else ym)

        def bw():
            if y.g is None:
                return
            gy = ops.as_map(y.g) if y.g.dim() == 2 else y.g
            if act != ACT_NONE:                          # fresh map, pad channels zero
                gz = self.act_bwd(gy, saved, act, slope, from_output=(act != ACT_GELU))
            elif p.N % 4 or ops.ld(gy) % 4 or gy.data_ptr() % 16:
                gz = _zero_padded(gy, p.N, self.device)  # the dgrad GEMM contracts over pad4(N) aligned channels: pad = 0
            else:
                gz = gy
            part = torch.empty(_wgrad_scratch(p), device=self.device)
            hip.call("ffsr_conv_wgrad_bf16x3" if ops.GEMM_MODE == "bf16x3" and WGRAD_BF16X3 else "ffsr_conv_wgrad_f32", _ptr(xm), ops.ld(xm), _ptr(gz), ops.ld(gz), _ptr(p.weight.g),
                     None if p.bias is None else _ptr(p.bias.g), _ptr(part),
                     part.numel(), B, H, W, p.Cin, p.N, p.KH, p.KW, p.pad, p.pad, _stream())
            if x.req:
                gzw = ops.widen(gz, p.bwd.Cin)
                gx = ops.conv2d(gzw, p.bwd,
                                tile_hint=0 if ops.thin3_ok(p.bwd, M, gzw) else conv_tile(M, p.Cin, p.KH * p.KW, p.N))
                self.acc(x, _unmap(gx) if is2d else gx)
        self._rec(bw)
        return y

    def gelu_conv_chain(self, x: Var, ps: Sequence[ConvP]) -> Var:
        """conv -> GELU -> conv -> GELU ... (the refine stack, enhanced_fusion_v2.py:569-576).  Where consecutive layers are wide
        3x3 layers (the planes tap-strip kernel's shapes) the activation between them lives as bf16 planes ONLY: the producing
        GEMM's epilogue writes the pre-activation z in fp32 (kept for GELU's backward) and GELU(z) as planes, which the next layer's
        forward GEMM and weight-gradient kernel both read -- no GELU pass, no split pass, no fp32 copy.  Other layers, other modes
        (FFSR_GEMM_MODE=f32, FFSR_PLANES=0): plain Tape.conv."""
        def planes_layer(p):
            return (ops.GEMM_MODE == "bf16x3" and ops.PLANES_AUTO and WGRAD_BF16X3 and p.KH == 3 and p.KW == 3 and p.pad == 1
                    and p.fwd.phi is not None and p.N % 128 == 0 and p.Cin % 128 == 0)
        B, H, W, _ = x.v.shape
        M = B * H * W
        if M < 65536 or not any(planes_layer(p) for p in ps):
            for p in ps:
                x = self.conv(x, p, act=ACT_GELU)
            return x
        i, n = 0, len(ps)
        while i < n:
            if not planes_layer(ps[i]):
                x = self.conv(x, ps[i], act=ACT_GELU)
                i += 1
                continue
            j = i
            while j < n and planes_layer(ps[j]):
                j += 1
            x = self._planes_run(x, ps[i:j])
            i = j
        return x

    def _planes_run(self, x: Var, ps: Sequence[ConvP]) -> Var:
        B, H, W, _ = x.v.shape
        xp = ops.split_planes(x.v)
        ins, zs = [], []
        for k, p in enumerate(ps):
            last = k == len(ps) - 1
            ins.append(xp)
            if last:       # the consumer of the run's output wants an fp32 map
                z = ops.conv2d(xp, p.fwd)
                y = ops.unary(z, act=ACT_GELU)
            else:
                z, xp = ops.conv2d(xp, p.fwd, act=ACT_GELU, out_planes=True, want_f32=True, pre_act_out=True)
            zs.append(z)
        out = Var(y)

        def bw():
            if out.g is None:
                return
            gy = out.g
            for k in range(len(ps) - 1, -1, -1):
                p = ps[k]
                # dZ = dY * GELU'(z) straight to planes: the input-gradient GEMM and the weight-gradient kernel both read them
                gz = ops.Planes(B, H, W, p.N, self.device)
                _, Mr, C, ldg = _mat(gy)
                hip.call("ffsr_act_bwd_planes_f32", _ptr(gy), ldg, _ptr(zs[k]), _mat(zs[k])[3], _ptr(gz.hi), _ptr(gz.lo), gz.Cp, Mr, C,
                         ACT_GELU, 0.0, 0, 1.0, _stream())
                part = torch.empty(_wgrad_scratch(p), device=self.device)
                hip.call("ffsr_conv_wgrad_bf16x3_planes", _ptr(ins[k].hi), _ptr(ins[k].lo), ins[k].Cp, None, 0, _ptr(gz.hi), _ptr(gz.lo),
                         gz.Cp, _ptr(p.weight.g), None if p.bias is None else _ptr(p.bias.g), _ptr(part), part.numel(), B, H, W,
                         p.Cin, p.N, p.KH, p.KW, p.pad, p.pad, _stream())
                if k > 0 or x.req:
                    gy = ops.conv2d(gz, p.bwd)
            if x.req:
                self.acc(x, gy)
        self._rec(bw)
        return out

    def linear(self, x: Var, p: ConvP, act=ACT_NONE) -> Var:
        return self.conv(x, p, act)

    def dwconv(self, x: Var, p: DwP) -> Var:
        y = Var(ops.dwconv2d(x.v, p.fwd))

        def bw():
            if y.g is None:
                return
            B, H, W, C = x.v.shape
            if p.param is not None:
                nchunk = max(1, min(1024, B * H * W // 512))
                part = torch.empty(nchunk * p.KH * p.KW * C, device=self.device)
                hip.call("ffsr_dwconv_wgrad_f32", _ptr(x.v), ops.ld(x.v), _ptr(y.g), ops.ld(y.g), _ptr(p.param.g), _ptr(part),
                         nchunk, B, H, W, C, p.KH, p.KW, p.ph, p.pw, _stream())
            if x.req:
                self.acc(x, ops.dwconv2d(y.g, p.bwd))
        self._rec(bw)
        return y

    # ------------------------------------------------------------------------------------------ normalisation
    def bn(self, x: Var, p: BnP, eps=1e-5, momentum=0.1) -> Var:
        """nn.BatchNorm2d in train mode: batch statistics, running statistics updated in place"""
        _, M, C, ldx = _mat(x.v)
        nchunk = max(1, min(1024, M // 256))
        part = torch.empty(nchunk * C, device=self.device)
        sums, stat, ss = (torch.empty(2 * C, device=self.device) for _ in range(3))
        hip.call("ffsr_bn_train_stats_f32", _ptr(x.v), ldx, M, C, _ptr(p.gamma.v), _ptr(p.beta.v), float(eps), float(momentum),
                 _ptr(part), nchunk, _ptr(sums), _ptr(stat), _ptr(ss), _ptr(p.run_mean), _ptr(p.run_var), _stream())
        p.calls += 1
        y = Var(ops.unary(x.v, cscale=ss[:C], cbias=ss[C:]))

        def bw():
            if y.g is None:
                return
            dx = ops._like(x.v)
            coef = torch.empty(3 * C, device=self.device)
            hip.call("ffsr_bn_train_bwd_f32", _ptr(x.v), ldx, _ptr(y.g), _mat(y.g)[3], _ptr(dx), _mat(dx)[3], M, C, _ptr(p.gamma.v),
                     _ptr(stat), _ptr(part), nchunk, _ptr(sums), _ptr(coef), _ptr(p.gamma.g), _ptr(p.beta.g), _stream())
            self.acc(x, dx)
        self._rec(bw)
        return y

    def layernorm(self, x: Var, gamma: Param, beta: Param, eps=1e-5) -> Var:
        y = Var(ops.layernorm(x.v, gamma.v, beta.v, eps))

        def bw():
            if y.g is None:
                return
            _, M, C, ldx = _mat(x.v)
            nblock = max(1, min(1024, M // 64))
            part = torch.empty(2 * nblock * C, device=self.device)
            dx = ops._like(x.v)
            hip.call("ffsr_layernorm_bwd_f32", _ptr(x.v), ldx, _ptr(gamma.v), float(eps), _ptr(y.g), _mat(y.g)[3], _ptr(dx),
                     _mat(dx)[3], _ptr(part), nblock, _ptr(gamma.g), _ptr(beta.g), M, C, _stream())
            self.acc(x, dx)
        self._rec(bw)
        return y

    # ------------------------------------------------------------------------------------------ elementwise
    def act(self, x: Var, act, slope=0.0, clamp=None, out=None) -> Var:
        """y = act(x); act ACT_CLAMP01 = clamp(x, 0, 1), ACT_CLAMP_MIN = clamp(x, min=slope)"""
        if act == ACT_CLAMP01:
            yv = ops.unary(x.v, clamp=(0.0, 1.0), out=out)
        elif act == ACT_CLAMP_MIN:
            yv = ops.unary(x.v, clamp=(float(slope), 3.0e38), out=out)
        else:
            yv = ops.unary(x.v, act=act, slope=slope, out=out)
        y = Var(yv, x.req)

        def bw():
            if y.g is not None and x.req:
                from_out = act in (ACT_RELU, ACT_LRELU, ACT_SIGMOID)
                self.acc(x, self.act_bwd(y.g, y.v if from_out else x.v, act, slope, from_out))
        self._rec(bw)
        return y

    def affine(self, x: Var, alpha: float, beta: float, out=None) -> Var:
        y = Var(ops.unary(x.v, alpha=alpha, beta=beta, out=out), x.req)

        def bw():
            if y.g is not None and x.req:
                self.acc(x, ops.unary(y.g, alpha=alpha))
        self._rec(bw)
        return y

    def copy(self, x: Var, out: torch.Tensor) -> Var:
        return self.affine(x, 1.0, 0.0, out=out)

    def add(self, a: Var, b: Var, alpha=1.0, beta=1.0, out=None) -> Var:
        y = Var(ops.scale_add(a.v, b.v, alpha=alpha, beta=beta, out=out), a.req or b.req)

        def bw():
            if y.g is None:
                return
            for v, s in ((a, alpha), (b, beta)):
                if v.req:
                    self.acc(v, ops.unary(y.g, alpha=s))
        self._rec(bw)
        return y

    def add_scaled(self, a: Optional[Var], f: Var, s: torch.Tensor, s_grad: Optional[torch.Tensor], beta=1.0, out=None) -> Var:
        """y = a + beta * s[0] * f with s a learnable DEVICE scalar (s_grad: where d loss / d s accumulates; None: constant)"""
        _, M, C, ldf = _mat(f.v)
        if out is None:
            out = ops._like(f.v)
        if a is None:
            hip.call("ffsr_axpby_dev_f32", _ptr(f.v), ldf, _ptr(s), float(beta), None, 0, None, 0.0, _ptr(out), _mat(out)[3], M, C,
                     _stream())
        else:
            hip.call("ffsr_axpby_dev_f32", _ptr(a.v), _mat(a.v)[3], None, 1.0, _ptr(f.v), ldf, _ptr(s), float(beta), _ptr(out),
                     _mat(out)[3], M, C, _stream())
        y = Var(out)

        def bw():
            if y.g is None:
                return
            if a is not None and a.req:
                self.acc(a, y.g, owned=False)
            if s_grad is not None:
                self.dot_into(s_grad, y.g, f.v, scale=beta)
            if f.req:
                gf = ops._like(f.v)
                hip.call("ffsr_axpby_dev_f32", _ptr(y.g), _mat(y.g)[3], _ptr(s), float(beta), None, 0, None, 0.0, _ptr(gf),
                         _mat(gf)[3], M, C, _stream())
                self.acc(f, gf)
        self._rec(bw)
        return y

    def mul(self, a: Var, b: Var, row_broadcast=False, alpha=1.0, out=None) -> Var:
        """y = alpha * a * b (b [.., 1] broadcast over the channels when row_broadcast)"""
        y = Var(ops.mul_add(a.v, b.v, row_broadcast=row_broadcast, alpha=alpha, out=out))

        def bw():
            if y.g is None:
                return
            if a.req:
                self.acc(a, ops.mul_add(y.g, b.v, row_broadcast=row_broadcast, alpha=alpha))
            if b.req:
                if row_broadcast:
                    _, M, C, ldg = _mat(y.g)
                    gb = ops._like(b.v)
                    hip.call("ffsr_rowdot_f32", _ptr(y.g), ldg, _ptr(a.v), _mat(a.v)[3], _ptr(gb), _mat(gb)[3], M, C, float(alpha),
                             0, _stream())
                    self.acc(b, gb)
                else:
                    self.acc(b, ops.mul_add(y.g, a.v, alpha=alpha))
        self._rec(bw)
        return y

    # ------------------------------------------------------------------------------------------ resamplers
    def bilinear(self, x: Var, Ho, Wo, out=None) -> Var:
        y = Var(ops.bilinear(x.v, Ho, Wo, out=out), x.req)

        def bw():
            if y.g is None or not x.req:
                return
            B, Hi, Wi, C = x.v.shape
            gx = ops.new_map(B, Hi, Wi, C, self.device)
            hip.call("ffsr_bilinear_bwd_f32", _ptr(y.g), ops.ld(y.g), _ptr(gx), ops.ld(gx), B, Hi, Wi, Ho, Wo, C, 1.0, 0, _stream())
            self.acc(x, gx)
        self._rec(bw)
        return y

    def avgpool2(self, x: Var) -> Var:
        y = Var(ops.avgpool2(x.v), x.req)

        def bw():
            if y.g is None or not x.req:
                return
            B, H, W, C = x.v.shape
            gx = ops.new_map(B, H, W, C, self.device)
            hip.call("ffsr_avgpool2_bwd_f32", _ptr(y.g), ops.ld(y.g), _ptr(gx), ops.ld(gx), B, H, W, C, 0, _stream())
            self.acc(x, gx)
        self._rec(bw)
        return y

    # ------------------------------------------------------------------------------------------ attention / fusion tails
    def pixel_mha(self, qkv: Var, S, T, E, heads, p_drop=0.0) -> Var:
        """nn.MultiheadAttention's core; p_drop > 0: attention dropout, every call of a step draws its own mask (seed =
        (tape seed, step, call index)) and the backward closure replays exactly that mask"""
        seed = 0
        if p_drop > 0:
            seed = (self.seed * 1000003 + self.step) * 4099 + self._draws
            self._draws += 1
        y = Var(ops.pixel_mha(qkv.v, S, T, E, heads, p_drop=p_drop, seed=seed))

        def bw():
            if y.g is None:
                return
            dq = torch.empty(S * T, 3 * E, device=self.device)
            scratch = torch.empty(2 * S * heads * T * T, device=self.device)
            hip.call("ffsr_pixel_mha_bwd_f32", _ptr(qkv.v), _mat(qkv.v)[3], _ptr(y.g), _mat(y.g)[3], _ptr(dq), 3 * E, _ptr(scratch),
                     float(p_drop), int(seed), S, T, E, heads, _stream())
            self.acc(qkv, dq)
        self._rec(bw)
        return y

    def softmax_c(self, x: Var) -> Var:
        _, M, C, ldx = _mat(x.v)
        yv = ops._like(x.v)
        hip.call("ffsr_softmax_c_f32", _ptr(x.v), ldx, _ptr(yv), _mat(yv)[3], M, C, _stream())
        y = Var(yv)

        def bw():
            if y.g is None:
                return
            dx = ops._like(x.v)
            hip.call("ffsr_softmax_c_bwd_f32", _ptr(yv), _mat(yv)[3], _ptr(y.g), _mat(y.g)[3], _ptr(dx), _mat(dx)[3], M, C, _stream())
            self.acc(x, dx)
        self._rec(bw)
        return y

    def expert_sum(self, x: Var, g: Var, normalize: bool) -> Var:
        """x [.., 12] = 4 experts x 3 channels, g [.., 4] -> [.., 3] (optionally divided by sum g + 1e-8)"""
        B, H, W, _ = x.v.shape
        M = B * H * W
        out = ops.new_map(B, H, W, 3, self.device)
        hip.call("ffsr_expert_sum_f32", _ptr(x.v), ops.ld(x.v), _ptr(g.v), ops.ld(g.v), _ptr(out), ops.ld(out), M, int(normalize),
                 _stream())
        y = Var(out)

        def bw():
            if y.g is None:
                return
            dx = torch.empty(B, H, W, 12, device=self.device)
            dg = torch.empty(B, H, W, 4, device=self.device)
            hip.call("ffsr_expert_sum_bwd_f32", _ptr(x.v), ops.ld(x.v), _ptr(g.v), ops.ld(g.v), _ptr(y.g), ops.ld(y.g), _ptr(dx), 12,
                     _ptr(dg), 4, M, int(normalize), 0, _stream())
            self.acc(x, dx)
            self.acc(g, dg)
        self._rec(bw)
        return y

    def selector_gates(self, raw: Var, diff: Var, temperature: Param) -> Var:
        y = Var(ops.selector_gates(raw.v, diff.v, temperature.v))

        def bw():
            if y.g is None:
                return
            B, h, w, _ = raw.v.shape
            draw = torch.empty(B, h, w, 4, device=self.device)
            ddiff = ops.new_map(B, h, w, 1, self.device)
            hip.call("ffsr_selector_gates_bwd_f32", _ptr(raw.v), ops.ld(raw.v), _ptr(diff.v), ops.ld(diff.v), _ptr(temperature.v),
                     _ptr(y.g), ops.ld(y.g), _ptr(draw), 4, _ptr(ddiff), ops.ld(ddiff), _ptr(self._partial()), N_PARTIAL,
                     _ptr(temperature.g), B * h * w, _stream())
            self.acc(raw, draw)
            self.acc(diff, ddiff)
        self._rec(bw)
        return y


# ---------------------------------------------------------------------------------------------- helpers
def _unmap(t: torch.Tensor) -> torch.Tensor:
    """[1,1,M,C] map -> [M,C] matrix view"""
    return t.as_strided((t.shape[2], t.shape[3]), (ops.ld(t), 1), t.storage_offset())


def _zero_padded(g: torch.Tensor, C: int, device) -> torch.Tensor:
    """a map whose channels C .. pad4(C)-1 are guaranteed zero (copy unless g already is the head of a zeroed buffer)"""
    B, H, W, _ = g.shape
    out = ops.new_map(B, H, W, C, device)
    return ops.unary(g, out=out)


def ch(c0: int, c1: int):
    """selector of the channel slice [c0, c1) of a map / matrix"""
    return lambda t: t[..., c0:c1]


def tok(e: int, T: int, B: int, h: int, w: int):
    """selector of token e of every pixel: rows e::T of a [B*h*w*T, E] matrix, as a [B,h,w,E] map"""
    def sel(t):
        ld_ = max(t.stride(0), t.shape[1])
        return t.as_strided((B, h, w, t.shape[1]), (h * w * T * ld_, w * T * ld_, T * ld_, 1), t.storage_offset() + e * ld_)
    return sel
