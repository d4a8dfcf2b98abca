"""Launch helpers: channels-last torch tensors in, HIP kernels through the C ABI, torch tensors out.

torch is used for device memory (allocation, views) and the current stream only; every arithmetic
operation goes through libffsr_hip.so.  A feature map is a tensor of shape [B, H, W, C] whose last
dimension is contiguous and whose pixel stride ("ld") may exceed C (channel slices of a wider buffer,
or channels padded to a multiple of 4).  Token matrices [M, C] are maps with B = H = 1.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import hip

ACT_NONE, ACT_GELU, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_SILU = range(6)

# GEMM arithmetic mode of conv2d/linear when no explicit tile_hint is given:
#   "f32"    exact f32-input MFMA (v_mfma_f32_32x32x2_f32), an fmaf chain like the reference's CPU path
#   "bf16x3" split-bf16 3-term MFMA (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~1e-5 relative per product, ~2x faster.
#            Default: full-depth parity vs the oracle is 1e-5 max-abs (PSNR 116 dB), 100x inside north_star's 1e-3
#            (tests/test_gpu_models.py::test_full_depth_experts_and_fusion_vs_oracle_64x64).  FFSR_GEMM_MODE=f32 selects
#            the exact kernel.
_ENV_MODE = os.environ.get("FFSR_GEMM_MODE", "bf16x3")
GEMM_MODE = "bf16x3" if _ENV_MODE == "bf16" else _ENV_MODE      # "bf16" = the bf16x3 kernels with one product term (hip.lib() sets it)


# Precision-budget experiment: FFSR_WEIGHT_LO=0 packs every weight with a ZERO lo plane, i.e. the split-bf16
# kernels then evaluate a_hi*w_hi + a_lo*w_hi -- the arithmetic of a 2-term ("bf16x2") mode with bf16-rounded weights -- without
# touching the kernels.  Measured by tools/precision_budget.py; not a default.
WEIGHT_LO = os.environ.get("FFSR_WEIGHT_LO", "1") != "0"


def set_weight_lo(on: bool):
    global WEIGHT_LO
    WEIGHT_LO = bool(on)


BN128_MIN_N = 1 << 30   # the 128-column tile (2 waves/SIMD) measured slower on every shape: keep BN = 64


GEMM_TERMS = 1 if _ENV_MODE == "bf16" else 3


def set_gemm_mode(mode: str):
    """"bf16x3": GEMMs, convs and DRCT's window attention on the split-bf16 MFMA; "f32": the exact kernels everywhere;
    "bf16": the bf16x3 kernels with ONE product term (plain bf16 operands, fp32 accumulate: BASELINE config 2's named precision;
    ~2e-2 max-abs on NAFNet alone -- a precision option, never the default)."""
    global GEMM_MODE, GEMM_TERMS
    assert mode in ("f32", "bf16x3", "bf16")
    GEMM_TERMS = 1 if mode == "bf16" else 3
    GEMM_MODE = "bf16x3" if mode == "bf16" else mode       # same kernels, same dispatch; only the product terms differ
    hip.call("ffsr_set_gemm_terms", GEMM_TERMS)


def gemm_mode_name() -> str:
    return "bf16" if (GEMM_MODE == "bf16x3" and GEMM_TERMS == 1) else GEMM_MODE


# bench.py sets this to a list to time every conv/GEMM launch with events on the launch stream:
# entries (start_event, end_event, algorithmic flops = 2 * M * N * KH*KW*Cin with the UNPADDED Cin, shape + kernel kind, bytes)
CONV_PROFILE = None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def ld(t: torch.Tensor) -> int:
    """Pixel stride of a channels-last map; validates the layout contract."""
    assert t.dtype == torch.float32 and t.is_cuda, "ffsr ops need fp32 device tensors"
    assert t.dim() == 4 and t.stride(3) == 1, f"not channels-last: {t.shape} {t.stride()}"
    B, H, W, C = t.shape
    s = t.stride(2)
    if s < C:                      # degenerate stride of a size-1 dimension
        assert W == 1, f"bad pixel stride: {t.shape} {t.stride()}"
        s = t.stride(1) if (H > 1 and t.stride(1) >= C) else (t.stride(0) if (B > 1 and t.stride(0) >= C) else C)
    if W > 1 and H > 1:
        assert t.stride(1) == W * s, f"rows not dense: {t.shape} {t.stride()}"
    if B > 1 and (H > 1 or W > 1):
        assert t.stride(0) == H * W * s, f"batches not dense: {t.shape} {t.stride()}"
    return s


def zeros(*shape, device) -> torch.Tensor:
    """fp32 device buffer cleared through the C ABI (ffsr_zero_f32 = hipMemsetAsync on the current stream): torch only
    allocates."""
    buf = torch.empty(*shape, device=device, dtype=torch.float32)
    if buf.numel():
        hip.call("ffsr_zero_f32", buf.data_ptr(), buf.numel(), _stream())
    return buf


def new_map(B, H, W, C, device, zero_pad=True) -> torch.Tensor:
    """[B,H,W,C] view of a buffer whose pixel stride is padded to a multiple of 4 (pad channels zeroed)."""
    Cp = pad4(C)
    if Cp == C:
        return torch.empty(B, H, W, C, device=device, dtype=torch.float32)
    if zero_pad:
        return zeros(B, H, W, Cp, device=device)[..., :C]
    return torch.empty(B, H, W, Cp, device=device, dtype=torch.float32)[..., :C]


def widen(t: torch.Tensor, C: int) -> torch.Tensor:
    """View of the same pixels with C channels (C <= ld): exposes the zero padding to a consumer."""
    s = ld(t)
    assert C <= s
    B, H, W, _ = t.shape
    return t.as_strided((B, H, W, C), (H * W * s, W * s, s, 1), t.storage_offset())


def as_map(x2d: torch.Tensor) -> torch.Tensor:
    """[M, C] (row stride ld) -> [1, 1, M, C] map."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    return x2d.as_strided((1, 1, x2d.shape[0], x2d.shape[1]), (0, 0, x2d.stride(0), 1), x2d.storage_offset())


def rows(t: torch.Tensor) -> int:
    return t.shape[0] * t.shape[1] * t.shape[2]


# ---------------------------------------------------------------------------------------------- split planes
def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class Planes:
    """A feature map stored PRE-SPLIT for the GEMM: bf16 hi / lo planes [2, B*H*W, Cp] (Cp = C rounded up to 32, pad
    channels zero) -- the same bytes as fp32, consumed by ffsr_conv2d_planes without any conversion work."""

    def __init__(self, B, H, W, C, device, buf=None):
        self.B, self.H, self.W, self.C = B, H, W, C
        self.Cp = pad32(C)
        self.buf = buf if buf is not None else torch.empty(2, B * H * W, self.Cp, device=device, dtype=torch.bfloat16)

    @property
    def shape(self):
        return (self.B, self.H, self.W, self.C)

    @property
    def M(self):
        return self.B * self.H * self.W

    @property
    def hi(self):
        return self.buf[0]

    @property
    def lo(self):
        return self.buf[1]

    def reshape_map(self, B, H, W):
        assert B * H * W == self.M
        return Planes(B, H, W, self.C, self.buf.device, self.buf)

    def to_f32(self) -> torch.Tensor:
        """hi + lo as an fp32 map (tests / debugging only)."""
        v = self.buf[0].float() + self.buf[1].float()
        return v[:, :self.C].reshape(self.B, self.H, self.W, self.C)


def split_planes(x: torch.Tensor, out: Optional[Planes] = None) -> Planes:
    """fp32 map / matrix -> Planes (one HBM pass)."""
    if x.dim() == 2:
        x = as_map(x)
    B, H, W, C = x.shape
    if out is None:
        out = Planes(B, H, W, C, x.device)
    hip.call("ffsr_split_planes", _ptr(x), ld(x), _ptr(out.hi), _ptr(out.lo), out.Cp, B * H * W, C, _stream())
    return out


# ---------------------------------------------------------------------------------------------- packed weights
@dataclass
class Conv:
    """A conv / linear layer packed for ffsr_conv2d_f32: wgt [N, KH*KW*Cin_pad] tap-major, channel-minor."""
    wgt: torch.Tensor
    bias: Optional[torch.Tensor]
    N: int
    Cin: int       # padded input channels the kernel contracts over
    KH: int = 1
    KW: int = 1
    stride: int = 1
    pad: int = 0
    Cin_true: int = 0   # unpadded input channels (algorithmic flop accounting)
    whi: Optional[torch.Tensor] = None   # bf16 hi plane [Npad, Kpad] (split-bf16 path)
    wlo: Optional[torch.Tensor] = None   # bf16 lo plane
    phi: Optional[torch.Tensor] = None   # bf16 hi plane [Npad768, taps * pad32(Cin)] (pre-split-input path)
    plo: Optional[torch.Tensor] = None
    Cp32: int = 0                        # channels per tap of phi / plo


def pack_conv(w: torch.Tensor, b: Optional[torch.Tensor], device, stride=1, pad=None, cin_pad=None,
              gate_pairs=False) -> Conv:
    """w: [N, Cin, KH, KW] (nn.Conv2d) or [N, K] (nn.Linear).  cin_pad: padded channel count of the input map.
    Load-time weight preparation (layout change, zero padding, bf16 hi/lo split): done with torch ops on `device`."""
    w = w.detach().float().to(device)
    if w.dim() == 2:
        w = w[:, :, None, None]
    if gate_pairs:
        # SimpleGate fused into the GEMM store (conv2d(..., gate=True)): output column q of the packed matrix is
        # x1[32 b + i] for q = 64 b + i and x2[32 b + i] for q = 64 b + 32 + i (i < 32), x1 / x2 = the channel halves
        c = w.shape[0] // 2
        assert w.shape[0] == 2 * c and c % 32 == 0
        q = torch.arange(2 * c, device=w.device)
        perm = (q // 64) * 32 + (q % 32) + (q % 64 >= 32) * c
        w = w[perm]
        b = None if b is None else b.detach().to(device)[perm]
    N, Cin, KH, KW = w.shape
    Cp = cin_pad if cin_pad is not None else pad4(Cin)
    assert Cp >= Cin and Cp % 4 == 0
    wt = w.permute(0, 2, 3, 1)                                             # [N, KH, KW, Cin]
    wp = torch.zeros(N, KH, KW, Cp, dtype=torch.float32, device=device)
    wp[..., :Cin] = wt
    if pad is None:
        pad = KH // 2
    w2 = wp.reshape(N, KH * KW * Cp).contiguous()
    # pre-split planes for the bf16x3 kernel: hi = bf16(w), lo = bf16(w - hi); rows padded to x128, K to x32
    Np, Kp = (N + 127) // 128 * 128, (KH * KW * Cp + 31) // 32 * 32
    hi = torch.zeros(Np, Kp, dtype=torch.bfloat16, device=device)
    lo = torch.zeros(Np, Kp, dtype=torch.bfloat16, device=device)
    h = w2.to(torch.bfloat16)
    hi[:N, :w2.shape[1]] = h
    if WEIGHT_LO:
        lo[:N, :w2.shape[1]] = (w2 - h.float()).to(torch.bfloat16)
    # planes path: every tap padded to a multiple of 32 channels, rows padded to x768 (any column tile of 64/128/192/256)
    C32, Np3 = pad32(Cin), (N + 767) // 768 * 768
    w3 = torch.zeros(Np3, KH, KW, C32, dtype=torch.float32, device=device)
    w3[:N, :, :, :Cin] = wt
    w3 = w3.reshape(Np3, KH * KW * C32)
    ph = w3.to(torch.bfloat16)
    pl = (w3 - ph.float()).to(torch.bfloat16) if WEIGHT_LO else torch.zeros_like(ph)
    return Conv(w2, None if b is None else b.detach().float().contiguous().to(device), N, Cp, KH, KW, stride,
                pad, Cin, hi, lo, ph, pl, C32)


_ZEROS = {}


def zero_page(device) -> torch.Tensor:
    key = str(device)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(64, device=device)   # 256 B, 16-byte aligned
    return _ZEROS[key]


GATE_FUSED = os.environ.get("FFSR_GATE_FUSED", "1") != "0"   # FFSR_GATE_FUSED=0: NAFNet's conv4 and SimpleGate stay separate (A/B runs)
STRIP3 = os.environ.get("FFSR_STRIP3", "1") != "0"   # FFSR_STRIP3=0: 3x3 convs stay on the per-tap tile kernel (A/B runs)


def planes_tile(M: int, N: int, K: int, conv3: bool = False):
    """(bm, bn, stages) of the planes GEMM (measured on MI355X, tools/planes_bench.py): least padded N weighted by the
    tile's efficiency (192 > 128 > 64 columns); 256-row tiles only pay on long-K shapes with many row tiles.
    conv3 (3x3, stride 1, pad 1): the tap-strip variant (stages 4; 64- / 128-column tiles) where it measured faster
    (tools/strip_bench.py: N = 45 / 60 at Cin 180: 200 / 214 -> 134 / 153 us; 128 -> 128 at HR: 2394 -> 2309 us; with
    N = 180 the 192-column per-tap tile stays ahead)."""
    if conv3 and STRIP3 and M >= 16384:
        if N <= 64:
            return 128, 64, 4
        if N <= 128 and K >= 512:
            return 128, 128, 4
    best = None
    for bn, eff in ((192, 1.0), (128, 0.95), (64, 0.8)):
        cost = (N + bn - 1) // bn * bn / eff
        if best is None or cost < best[0]:
            best = (cost, bn)
    bn = best[1]
    if bn == 128 and K >= 512 and M >= 256 * 1024:
        return 256, 128, 3
    return 128, bn, 2


PLANES_ACTS = (ACT_NONE, ACT_GELU, ACT_RELU, ACT_LRELU)
THIN3 = os.environ.get("FFSR_THIN3", "1") != "0"   # FFSR_THIN3=0: the N <= 4 3x3 heads stay on the GEMM kernels (A/B runs)


def thin3_ok(cv: "Conv", M: int, x: Optional[torch.Tensor] = None) -> bool:
    """3x3 / stride 1 / pad 1 with N <= 4 over an fp32 map: ffsr_conv3x3_thin_f32 (a streaming fp32 reduction) instead of a
    32-column MFMA tile that is 7/8 padding.  x (the input map, if known): the kernel reads 16-byte vectors, so the pixel
    stride must be a multiple of 4 and the first pixel 16-byte aligned -- a channel-slice view at an odd offset takes the
    GEMM kernels instead."""
    ok = (THIN3 and cv.KH == 3 and cv.KW == 3 and cv.stride == 1 and cv.pad == 1 and cv.N <= 4
          and cv.Cin in (8, 16, 32, 64, 128) and cv.N <= cv.Cin // 4 and M >= 4096)
    if ok and x is not None:
        ok = ld(x) % 4 == 0 and x.data_ptr() % 16 == 0 and ld(x) >= cv.Cin and rows(x) * ld(x) * 4 < 2 ** 32
    return ok
PLANES_AUTO = os.environ.get("FFSR_PLANES", "1") != "0"   # FFSR_PLANES=0: every GEMM takes its fp32 input directly


def conv2d(x, cv: Conv, *, act=ACT_NONE, slope=0.0, out: Optional[torch.Tensor] = None,
           res: Optional[torch.Tensor] = None, cvec=None, rvec=None, cscale=1.0, rscale=1.0, shuffle=0,
           akscale: Optional[torch.Tensor] = None, tile_hint=0, out_planes=None, want_f32=True, bm=0, bn=0, stages=0,
           gate=False, pre_act_out=False):
    """x [B,H,W,>=Cin] fp32 map or Planes -> [B,Ho,Wo,N] (or [B,2Ho,2Wo,N/4] with shuffle=2).
    gate=True (split-bf16 mode, fp32 input, weights packed with gate_pairs=True): the store multiplies the two channel
    halves (NAFNet's SimpleGate) -> [B,Ho,Wo,N/2]; res / cvec / rvec refer to the N/2 output channels.
    out_planes: True / a Planes object -> also emit the result as bf16 hi / lo planes (returned as (out, planes), or
    only the planes when want_f32 is False).
    pre_act_out (Planes input, both outputs): the fp32 map receives the pre-activation z, the planes act(z)."""
    B, H, W, _ = x.shape
    Ho = (H + 2 * cv.pad - cv.KH) // cv.stride + 1
    Wo = (W + 2 * cv.pad - cv.KW) // cv.stride + 1
    if shuffle:
        oshape = (B, 2 * Ho, 2 * Wo, cv.N // 4)
    elif gate:
        oshape = (B, Ho, Wo, cv.N // 2)
    else:
        oshape = (B, Ho, Wo, cv.N)
    is_planes = isinstance(x, Planes)
    if (not is_planes and not gate and not shuffle and akscale is None and cvec is None and rvec is None and out_planes is None
            and tile_hint == 0 and thin3_ok(cv, B * Ho * Wo, x)):
        if out is None:
            out = new_map(*oshape, x.device)
        assert tuple(out.shape) == oshape and (res is None or tuple(res.shape) == oshape)
        hip.call("ffsr_conv3x3_thin_f32", _ptr(x), ld(x), _ptr(cv.wgt), _ptr(cv.bias), _ptr(out), ld(out), _ptr(res),
                 0 if res is None else ld(res), B, H, W, cv.Cin, cv.N, act, float(slope), float(cscale), float(rscale), _stream())
        return out
    if gate:
        if is_planes or GEMM_MODE != "bf16x3" or shuffle or act != ACT_NONE or cv.N % 64 or B * Ho * Wo <= 64 * 24:
            raise ValueError("gate=True needs an fp32 map input, the split-bf16 mode, no activation and N % 64 == 0")
        shuffle, tile_hint = 3, 128           # store mode 3 of ffsr_conv2d_bf16x3
    if (not is_planes and PLANES_AUTO and GEMM_MODE == "bf16x3" and not shuffle and akscale is None and tile_hint == 0
            and act in PLANES_ACTS and cv.KH * cv.KW > 1 and cv.stride == 1 and cv.Cin_true >= 32 and cv.N > 64 and B * Ho * Wo >= 16384):
        # long-K conv on an fp32 map: one extra split pass (read + write of the input) buys the ~1.5x faster planes GEMM
        x = split_planes(widen(x, cv.Cin_true) if x.shape[3] != cv.Cin_true else x)
        is_planes = True
    if out_planes is True:
        out_planes = Planes(B, Ho, Wo, cv.N, x.buf.device if is_planes else x.device)
    if out_planes is not None and not is_planes:
        raise ValueError("planes output needs a Planes input (split_planes the input first)")
    dev_ = x.buf.device if is_planes else x.device
    if out is None and (want_f32 or out_planes is None):
        out = new_map(*oshape, dev_)
    if out is not None:
        assert tuple(out.shape) == oshape, f"{tuple(out.shape)} != {oshape}"
    ldr = 0
    if res is not None:
        assert tuple(res.shape) == oshape
        ldr = ld(res)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    kernel_kind = 0      # for CONV_PROFILE: 0 fp32-input kernel, 1 planes tile kernel, 2 planes tap-strip kernel
    if is_planes:
        assert x.Cp == cv.Cp32 and x.C == cv.Cin_true, f"planes input {x.shape} (Cp {x.Cp}) vs conv Cin {cv.Cin_true}"
        assert not shuffle and akscale is None and act in PLANES_ACTS
        if out_planes is not None:
            assert out_planes.shape == oshape
        tile = (bm or 128, bn, stages) if bn else planes_tile(B * Ho * Wo, cv.N, cv.KH * cv.KW * x.Cp,
                                                              cv.KH == 3 and cv.KW == 3 and cv.stride == 1 and cv.pad == 1)
        kernel_kind = 2 if tile[2] == 4 else 1
        hip.call("ffsr_conv2d_planes", _ptr(x.hi), _ptr(x.lo), x.Cp, _ptr(cv.phi), _ptr(cv.plo), cv.phi.shape[0],
                 _ptr(zero_page(dev_)), _ptr(cv.bias), _ptr(out), _ptr(res), _ptr(cvec), _ptr(rvec),
                 None if out_planes is None else _ptr(out_planes.hi), None if out_planes is None else _ptr(out_planes.lo),
                 0 if out_planes is None else out_planes.Cp, B, H, W, cv.N, 0 if out is None else ld(out), ldr, cv.KH,
                 cv.KW, cv.stride, cv.pad, cv.pad, act | (0x100 if pre_act_out else 0), float(slope), float(cscale), float(rscale),
                 *tile, _stream())
    else:
        if pre_act_out:
            raise ValueError("pre_act_out needs a Planes input")
        ldi = ld(x)
        assert ldi >= cv.Cin, f"input stride {ldi} < packed Cin {cv.Cin}"
        akrows = 0
        if akscale is not None:
            assert akscale.shape == (B, cv.Cin) and akscale.is_contiguous()
            akrows = H * W
        if tile_hint == 0 and GEMM_MODE == "bf16x3" and B * Ho * Wo > 64 * 24 and cv.KH * cv.KW <= 32:
            tile_hint = 32 if cv.N <= 32 else (64 if cv.N <= 64 or BN128_MIN_N > cv.N else 128)
        if tile_hint in (32, 64, 128):
            hip.call("ffsr_conv2d_bf16x3", _ptr(x), _ptr(cv.whi), _ptr(cv.wlo), cv.whi.shape[1], cv.whi.shape[0],
                     _ptr(zero_page(x.device)), _ptr(cv.bias), _ptr(out), _ptr(res), _ptr(cvec), _ptr(rvec), _ptr(akscale),
                     B, H, W, cv.Cin, ldi, cv.N, ld(out), ldr, cv.KH, cv.KW, cv.stride, cv.pad, cv.pad, act, float(slope),
                     float(cscale), float(rscale), shuffle, akrows, tile_hint, _stream())
        else:
            hip.call("ffsr_conv2d_f32", _ptr(x), _ptr(cv.wgt), _ptr(cv.bias), _ptr(out), _ptr(res), _ptr(cvec), _ptr(rvec),
                     _ptr(akscale), B, H, W, cv.Cin, ldi, cv.N, ld(out), ldr, cv.KH, cv.KW, cv.stride, cv.pad, cv.pad,
                     act, float(slope), float(cscale), float(rscale), shuffle, akrows, tile_hint, _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * B * Ho * Wo * cv.N * cv.KH * cv.KW * cv.Cin_true,
                     (B * Ho * Wo, cv.N, cv.KH * cv.KW * cv.Cin, cv.KH, kernel_kind),
                     4.0 * (B * H * W * cv.Cin_true + cv.N * cv.KH * cv.KW * cv.Cin_true
                            + B * Ho * Wo * cv.N * (2 if res is not None else 1))))
    if out_planes is not None:
        return (out, out_planes) if out is not None else out_planes
    return out


def linear(x2d, cv: Conv, **kw):
    """x2d [M, K] matrix or Planes -> [M, N]; keyword tensors (out / res) are [M, *] matrices too."""
    for k in ("out", "res"):
        if kw.get(k) is not None:
            kw[k] = as_map(kw[k])
    y = conv2d(x2d if isinstance(x2d, Planes) else as_map(x2d), cv, **kw)

    def mat(t):
        return t.as_strided((t.shape[2], t.shape[3]), (ld(t), 1), t.storage_offset())

    if isinstance(y, tuple):
        return mat(y[0]), y[1]
    return y if isinstance(y, Planes) else mat(y)


# ---------------------------------------------------------------------------------------------- token-stationary chains
@dataclass
class TokChain:
    """Weights of a per-token chain  W2 . act(W1 . pre(x) + b1) + b2  packed for ffsr_tok_chain_f32 (fragment-major bf16
    hi / lo: every 1 KB piece is one wave's MFMA A fragment, lane-linear)."""
    w1: torch.Tensor          # bf16 [steps, G, KS1, 2, 64, 8]
    b1: torch.Tensor          # fp32 [steps * G * 16]
    w2: torch.Tensor          # bf16 [steps, NT2, 2, 64, 8]
    b2: Optional[torch.Tensor]
    K: int
    N: int
    H: int                    # true hidden width (algorithmic flops)
    steps: int
    mode: int                 # 0: GELU MLP, 1: SimpleGate
    pre_ln: bool
    eps1: float


TOK_SHAPES = {0: ((4, 8), (6, 12), (7, 14), (8, 16), (9, 18), (10, 20)), 1: ((2, 4), (4, 8))}


# k-step counts ffsr_tok_head_chain_f32 is instantiated for: mode 0 (x1 + mlp(norm(x1))), mode 1 (gated chain), "post" (mode 0 with
# the post-LayerNorm form norm(mlp(x1)) + x1)
TOK_HEAD_KS = {0: (6, 7, 8, 9, 10), 1: (2, 4), "post": (6,)}


def tok_head_ok(K: int, mode) -> bool:
    return K % 4 == 0 and (K + 31) // 32 in TOK_HEAD_KS[mode]


def tok_chain_ok(K: int, N: int, mode: int) -> bool:
    return K % 4 == 0 and N % 4 == 0 and ((K + 31) // 32, (N + 31) // 32 * 2) in TOK_SHAPES[mode]


def _split_bf16(w: torch.Tensor):
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16) if WEIGHT_LO else torch.zeros_like(hi)
    return hi, lo


def pack_tok_chain(w1, b1, w2, b2, device, *, mode=0, ln=None, eps=1e-5) -> TokChain:
    """w1 [H, K] (+ b1 [H]), w2 [N, H'] (+ b2 [N]) as nn.Linear / 1x1 nn.Conv2d weights; H' = H (mode 0) or H / 2 (mode 1,
    SimpleGate: hidden rows 0..H/2-1 times rows H/2..H-1).  ln = (gamma, beta): the chain starts with nn.LayerNorm(K); its
    affine part is folded into the first layer (W1' = W1 diag(gamma), b1' = b1 + W1 beta), the kernel only normalises.
    Load-time weight preparation with torch ops on `device`."""
    w1 = w1.detach().float().to(device).reshape(w1.shape[0], -1)
    w2 = w2.detach().float().to(device).reshape(w2.shape[0], -1)
    H, K = w1.shape
    N, H2 = w2.shape
    b1 = torch.zeros(H, device=device) if b1 is None else b1.detach().float().to(device).reshape(-1)
    if ln is not None:
        gamma, beta = (t.detach().float().to(device).reshape(-1) for t in ln)
        b1 = b1 + w1 @ beta
        w1 = w1 * gamma[None, :]
    assert tok_chain_ok(K, N, mode), (K, N, mode)
    KS1, NT2 = (K + 31) // 32, (N + 31) // 32 * 2
    Hg = H2                                      # width of the second GEMM's contraction
    assert Hg == (H // 2 if mode == 1 else H)
    steps = (Hg + 31) // 32
    Hp = steps * 32

    # ---- first layer: rows in tile order.  mode 0: natural; mode 1: per step [x1 a, x1 b, x2 a, x2 b]
    def rows_padded(w, b, n):
        wp = torch.zeros(Hp, KS1 * 32, device=device)
        bp = torch.zeros(Hp, device=device)
        wp[:n, :K] = w
        bp[:n] = b
        return wp, bp

    if mode == 0:
        wp, bp = rows_padded(w1, b1, H)
        wt = wp.reshape(steps, 2, 16, KS1 * 32)
        bt = bp.reshape(steps, 2, 16)
    else:
        wa, ba = rows_padded(w1[:Hg], b1[:Hg], Hg)
        wb, bb = rows_padded(w1[Hg:], b1[Hg:], Hg)
        wt = torch.cat([wa.reshape(steps, 2, 16, KS1 * 32), wb.reshape(steps, 2, 16, KS1 * 32)], 1)   # [steps, 4, 16, Kp]
        bt = torch.cat([ba.reshape(steps, 2, 16), bb.reshape(steps, 2, 16)], 1)
    G = wt.shape[1]
    # A fragment of tile (st, g), k step s: lane (q = l >> 4, r = l & 15) holds W[16 tile + r][32 s + 8 q + j]
    f1 = wt.reshape(steps, G, 16, KS1, 4, 8).permute(0, 1, 3, 4, 2, 5).reshape(steps, G, KS1, 64, 8)
    h1, l1 = _split_bf16(f1)
    w1p = torch.stack([h1, l1], 3).contiguous()                              # [steps, G, KS1, 2, 64, 8]
    # ---- second layer: tile nt, step st: lane (q, r) element j holds W2[16 nt + r][32 st + 16 (j >> 2) + 4 q + (j & 3)]
    # (the order in which the first GEMM's accumulators hold the hidden features of a token)
    # rows in the LANE-COLUMN order: tile nt, row r holds output feature 32 (nt >> 1) + 8 (r >> 2) + 4 (nt & 1) + (r & 3), so
    # that lane q of the kernel ends up with the output columns 32 s + 8 q + 0..7 -- the columns of the input row it loaded
    w2n = torch.zeros(NT2 * 16, Hp, device=device)
    w2n[:N, :Hg] = w2
    nt_ = torch.arange(NT2, device=device)[:, None]
    r_ = torch.arange(16, device=device)[None, :]
    w2p_ = w2n[(32 * (nt_ // 2) + 8 * (r_ // 4) + 4 * (nt_ % 2) + (r_ % 4)).reshape(-1)]
    f2 = w2p_.reshape(NT2, 16, steps, 2, 4, 4).permute(2, 0, 4, 1, 3, 5).reshape(steps, NT2, 64, 8)
    h2, l2 = _split_bf16(f2)
    w2p = torch.stack([h2, l2], 2).contiguous()                              # [steps, NT2, 2, 64, 8]
    b2p = None
    if b2 is not None:
        b2p = torch.zeros(NT2 * 16, device=device)
        b2p[:N] = b2.detach().float().to(device).reshape(-1)
    return TokChain(w1p, bt.reshape(-1).contiguous(), w2p, b2p, K, N, H, steps, mode, ln is not None, float(eps))


@dataclass
class TokGemm:
    """A linear layer (optionally preceded by a LayerNorm) packed for ffsr_tok_gemm_f32."""
    w1: torch.Tensor          # bf16 [steps, 2, KS1, 2, 64, 8]
    b1: torch.Tensor          # fp32 [steps * 32]
    K: int
    N: int
    pre_ln: bool
    eps1: float


TOK_GEMM_KS = (2, 4, 6, 7, 8, 9, 10, 12)


def tok_gemm_ok(K: int, N: int) -> bool:
    return K % 4 == 0 and N % 4 == 0 and (K + 31) // 32 in TOK_GEMM_KS


def pack_tok_gemm(w, b, device, *, ln=None, eps=1e-5, check=True) -> TokGemm:
    """w [N, K] (+ b [N]) as an nn.Linear / 1x1 nn.Conv2d weight; ln = (gamma, beta) of a LayerNorm(K) in front of it (folded:
    W' = W diag(gamma), b' = b + W beta; the kernel only normalises)."""
    w = w.detach().float().to(device).reshape(w.shape[0], -1)
    N, K = w.shape
    b = torch.zeros(N, device=device) if b is None else b.detach().float().to(device).reshape(-1)
    if ln is not None:
        gamma, beta = (t.detach().float().to(device).reshape(-1) for t in ln)
        b = b + w @ beta
        w = w * gamma[None, :]
    assert not check or tok_gemm_ok(K, N), (K, N)
    KS1, steps = (K + 31) // 32, (N + 31) // 32
    wn = torch.zeros(steps * 32, KS1 * 32, device=device)
    bn = torch.zeros(steps * 32, device=device)
    wn[:N, :K] = w
    bn[:N] = b
    # rows in the lane-column order: step st, tile g, row r holds output feature 32 st + 8 (r >> 2) + 4 g + (r & 3)
    st_ = torch.arange(steps, device=device)[:, None, None]
    g_ = torch.arange(2, device=device)[None, :, None]
    r_ = torch.arange(16, device=device)[None, None, :]
    perm = (32 * st_ + 8 * (r_ // 4) + 4 * g_ + (r_ % 4)).reshape(-1)
    wt, bt = wn[perm].reshape(steps, 2, 16, KS1 * 32), bn[perm]
    f1 = wt.reshape(steps, 2, 16, KS1, 4, 8).permute(0, 1, 3, 4, 2, 5).reshape(steps, 2, KS1, 64, 8)
    h1, l1 = _split_bf16(f1)
    return TokGemm(torch.stack([h1, l1], 3).contiguous(), bt.contiguous(), K, N, ln is not None, float(eps))


def tok_gemm(x2d, tg: TokGemm, *, act=ACT_NONE, slope=0.0, cvec=None, cscale=1.0, out=None, out_planes=None, want_f32=True):
    """x2d [M, >=K] fp32 rows -> act(W pre(x) + b) * cvec * cscale as [M, N] fp32 and / or planes (see tok_chain)."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.shape[1] >= tg.K
    M = x2d.shape[0]
    if out_planes is True:
        out_planes = Planes(1, 1, M, tg.N, x2d.device)
    if out is None and (want_f32 or out_planes is None):
        out = torch.empty(M, tg.N, device=x2d.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    hip.call("ffsr_tok_gemm_f32", _ptr(x2d), x2d.stride(0), _ptr(tg.w1), _ptr(tg.b1), _ptr(cvec), _ptr(out),
             0 if out is None else out.stride(0), None if out_planes is None else _ptr(out_planes.hi),
             None if out_planes is None else _ptr(out_planes.lo), 0 if out_planes is None else out_planes.Cp, M, tg.K, tg.N,
             int(tg.pre_ln), tg.eps1, act, float(slope), float(cscale), tok_waves(M, (tg.K + 31) // 32), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * tg.K * tg.N, (M, tg.N, tg.K, 1, 3), 4.0 * (M * tg.K + tg.N * tg.K + M * tg.N)))
    if out_planes is not None:
        return (out, out_planes) if out is not None else out_planes
    return out


TOK_WAVES = int(os.environ.get("FFSR_TOK_WAVES", "0"))
TOK_FUSED = os.environ.get("FFSR_TOK", "1") != "0"   # FFSR_TOK=0: LayerNorm / fc1 / fc2 stay separate launches (A/B runs)


HEAD_FUSED = os.environ.get("FFSR_TOK_HEAD", "1") != "0"   # FFSR_TOK_HEAD=0: attention proj (+ residual / LayerNorm) as own launches
# MambaIR's out_norm + gate + out_proj + skip + ln_2 as ONE kernel (ffsr_tok_proj_f32, 4-wave workgroups); FFSR_TOK_PROJ=0: the three
# launches it replaces.  History of the measurement (tools/proj_bench.py, 352 x 512 x 360): first version 594 us against 591 us for
# the three launches (its epilogue waited for every bias / LayerNorm-vector / residual load separately, one exec-masked block per
# tile); with the batched epilogue loads 492 us against 556 us.
PROJ_FUSED = os.environ.get("FFSR_TOK_PROJ", "1") != "0"


def tok_enabled() -> bool:
    """the fused token chains replace launches of the split-bf16 mode only (the exact f32 mode keeps the f32-MFMA GEMMs)"""
    return TOK_FUSED and GEMM_MODE == "bf16x3"


def tok_waves(M: int, KS1: int) -> int:
    """waves (of 16 tokens) per persistent workgroup: 8 (two per SIMD); 4 when that leaves most of the chip's CUs without a tile"""
    if TOK_WAVES:
        return TOK_WAVES
    return 8 if M >= 128 * 256 else 4


def tok_chain(x2d, tc: TokChain, *, res=None, rscale=1.0, rvec=None, cvec=None, cscale=1.0, post_ln=None, eps2=1e-5,
              res2=None, out=None, out_planes=None, want_f32=True, tail=None):
    """x2d [M, >=K] fp32 rows -> [M, N]:  y = (W2 act(W1 pre(x) + b1) + b2) * cvec * cscale + res * rvec * rscale;
    post_ln = (gamma, beta): out = LayerNorm(y) * gamma + beta (+ res2).  out_planes: True / a Planes -> also emit bf16
    hi / lo planes (returns (out, planes), or only the planes when want_f32 is False).
    tail = dict(tg=TokGemm, out=[M, N3] view, act=, slope=, cscale=, res=, rscale=): a third linear layer on y inside the same
    kernel (out3 = act(W3 y + b3) * cscale + res * rscale); y itself is only stored when want_f32 -- returns out3 (or (y, out3))."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.shape[1] >= tc.K
    M = x2d.shape[0]
    if tail is not None:
        tg, out3 = tail["tg"], tail["out"]
        assert cvec is None and rvec is None and post_ln is None and out_planes is None and cscale == 1.0 and rscale == 1.0
        assert tg.K == tc.N and not tg.pre_ln and out3.shape == (M, tg.N) and out3.stride(1) == 1
        if out is None and want_f32:
            out = torch.empty(M, tc.N, device=x2d.device)
        r3 = tail.get("res")
        prof = CONV_PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.call("ffsr_tok_chain_tail_f32", _ptr(x2d), x2d.stride(0), _ptr(tc.w1), _ptr(tc.b1), _ptr(tc.w2), _ptr(tc.b2), _ptr(res),
                 0 if res is None else res.stride(0), _ptr(out), 0 if out is None else out.stride(0), M, tc.K, tc.N, tc.steps, tc.mode,
                 int(tc.pre_ln), tc.eps1, _ptr(tg.w1), _ptr(tg.b1), _ptr(r3), 0 if r3 is None else r3.stride(0), _ptr(out3),
                 out3.stride(0), tg.N, tail.get("act", ACT_NONE), float(tail.get("slope", 0.0)), float(tail.get("cscale", 1.0)),
                 float(tail.get("rscale", 1.0)), tok_waves(M, (tc.K + 31) // 32), _stream())
        if prof is not None:
            e1.record()
            prof.append((e0, e1, 2.0 * M * (tc.K * tc.H + tc.H * tc.N + tc.N * tg.N), (M, tc.N, tc.K, 1, 3),
                         4.0 * (M * tc.K + tc.H * tc.K + tc.N * tc.H + tg.N * tc.N + M * tg.N * (1 + (r3 is not None))
                                + (M * tc.N if out is not None else 0))))
        return out3 if out is None else (out, out3)
    if out_planes is True:
        out_planes = Planes(1, 1, M, tc.N, x2d.device)
    if out is None and (want_f32 or out_planes is None):
        out = torch.empty(M, tc.N, device=x2d.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    g2, be2 = post_ln if post_ln is not None else (None, None)
    hip.call("ffsr_tok_chain_f32", _ptr(x2d), x2d.stride(0), _ptr(tc.w1), _ptr(tc.b1), _ptr(tc.w2), _ptr(tc.b2), _ptr(cvec),
             _ptr(res), 0 if res is None else res.stride(0), _ptr(rvec), _ptr(g2), _ptr(be2), _ptr(res2),
             0 if res2 is None else res2.stride(0), _ptr(out), 0 if out is None else out.stride(0),
             None if out_planes is None else _ptr(out_planes.hi), None if out_planes is None else _ptr(out_planes.lo),
             0 if out_planes is None else out_planes.Cp, M, tc.K, tc.N, tc.steps, tc.mode, int(tc.pre_ln), tc.eps1, float(eps2),
             float(cscale), float(rscale), tok_waves(M, (tc.K + 31) // 32), _stream())
    if prof is not None:
        e1.record()
        Hc = tc.H if tc.mode == 0 else tc.H // 2
        prof.append((e0, e1, 2.0 * M * (tc.K * tc.H + Hc * tc.N), (M, tc.N, tc.K, 1, 3),
                     4.0 * (M * tc.K + tc.H * tc.K + tc.N * Hc + M * tc.N * (1 + (res is not None) + (res2 is not None)))))
    if out_planes is not None:
        return (out, out_planes) if out is not None else out_planes
    return out


def tok_head_chain(a2d, head: TokGemm, tc: TokChain, *, in_scale=None, head_ln=None, eps0=1e-5, hres=None, hres2=None, hvec2=None,
                   rows_per_batch=0, post_ln=None, eps2=1e-5, out=None, out_planes=None, want_f32=True, tail=None):
    """a2d [M, K] -> the chain `tc` applied to x1 = LN0?(head(a)) + hres + hres2 * hvec2[batch], one kernel, x1 never stored
    (ffsr_tok_head_chain_f32).  post_ln None: y = x1 + mlp(pre(x1)); (gamma, beta): y = LN(mlp(x1)) + x1.
    Returns like tok_chain (out / planes / tail output)."""
    assert a2d.dim() == 2 and a2d.stride(1) == 1 and head.K == head.N == tc.K == tc.N and not head.pre_ln and tc.mode in (0, 1)
    assert in_scale is None or (in_scale.is_contiguous() and in_scale.shape[-1] == head.K and rows_per_batch > 0)
    M = a2d.shape[0]
    out3 = None
    if tail is not None:
        tg, out3 = tail["tg"], tail["out"]
        assert tg.K == tc.N and not tg.pre_ln and out3.shape == (M, tg.N) and out3.stride(1) == 1 and out_planes is None
    if out_planes is True:
        out_planes = Planes(1, 1, M, tc.N, a2d.device)
    if out is None and want_f32:
        out = torch.empty(M, tc.N, device=a2d.device)
    g0, be0 = head_ln if head_ln is not None else (None, None)
    g2, be2 = post_ln if post_ln is not None else (None, None)
    r3 = None if tail is None else tail.get("res")
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    hip.call("ffsr_tok_head_chain_f32", _ptr(a2d), a2d.stride(0), _ptr(in_scale), _ptr(head.w1), _ptr(head.b1), _ptr(g0), _ptr(be0), float(eps0),
             _ptr(hres), 0 if hres is None else hres.stride(0), _ptr(hres2), 0 if hres2 is None else hres2.stride(0), _ptr(hvec2),
             int(rows_per_batch), _ptr(tc.w1), _ptr(tc.b1), _ptr(tc.w2), _ptr(tc.b2), _ptr(g2), _ptr(be2), float(eps2), _ptr(out),
             0 if out is None else out.stride(0), None if out_planes is None else _ptr(out_planes.hi),
             None if out_planes is None else _ptr(out_planes.lo), 0 if out_planes is None else out_planes.Cp, M, tc.K, tc.steps,
             tc.mode, int(tc.pre_ln), tc.eps1, None if tail is None else _ptr(tail["tg"].w1), None if tail is None else _ptr(tail["tg"].b1),
             _ptr(r3), 0 if r3 is None else r3.stride(0), _ptr(out3), 0 if out3 is None else out3.stride(0),
             0 if tail is None else tail["tg"].N, ACT_NONE if tail is None else tail.get("act", ACT_NONE),
             0.0 if tail is None else float(tail.get("slope", 0.0)), 1.0 if tail is None else float(tail.get("cscale", 1.0)),
             1.0 if tail is None else float(tail.get("rscale", 1.0)), tok_waves(M, (tc.K + 31) // 32), _stream())
    if prof is not None:
        e1.record()
        n3 = 0 if tail is None else tail["tg"].N
        prof.append((e0, e1, 2.0 * M * (tc.K * tc.K + tc.K * tc.H + tc.H * tc.N + tc.N * n3), (M, tc.N, tc.K, 1, 3),
                     4.0 * (M * tc.K * (1 + (hres is not None) + (hres2 is not None)) + tc.K * tc.K + 2 * tc.H * tc.K + n3 * tc.N
                            + M * n3 * (1 + (r3 is not None)) + (M * tc.N if out is not None else 0))))
    if tail is not None:
        return out3 if out is None else (out, out3)
    if out_planes is not None:
        return (out, out_planes) if out is not None else out_planes
    return out


def tok_proj_ok(K: int, N: int) -> bool:
    return K % 4 == 0 and N % 4 == 0 and (K + 31) // 32 == 12 and (N + 31) // 32 == 6


def tok_proj(x2d, tg: TokGemm, *, xdirs=1, xstride=0, z=None, pro_ln=None, peps=1e-5, cvec=None, cscale=1.0, res=None, rvec=None,
             rscale=1.0, post_ln=None, eps2=1e-5, out_pre_ln=False, out=None, out_planes=None, want_f32=True, waves=None):
    """ffsr_tok_proj_f32: y = (W0 a + b0) * cvec * cscale + res * rvec * rscale with a = [sum of xdirs partial rows ->]
    [LayerNorm(pro_ln) ->] [* silu(z)]; post_ln: planes (and out unless out_pre_ln) receive LayerNorm(y).
    x2d [M, >= K] is the first partial input; the others lie xstride elements apart."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.shape[1] >= tg.K and not tg.pre_ln
    M = x2d.shape[0]
    if out_planes is True:
        out_planes = Planes(1, 1, M, tg.N, x2d.device)
    if out is None and (want_f32 or out_planes is None):
        out = torch.empty(M, tg.N, device=x2d.device)
    pg, pb = pro_ln if pro_ln is not None else (None, None)
    g2, be2 = post_ln if post_ln is not None else (None, None)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    hip.call("ffsr_tok_proj_f32", _ptr(x2d), int(xstride), int(xdirs), x2d.stride(0), _ptr(z), 0 if z is None else z.stride(0),
             _ptr(pg), _ptr(pb), float(peps), _ptr(tg.w1), _ptr(tg.b1), _ptr(cvec), _ptr(res), 0 if res is None else res.stride(0),
             _ptr(rvec), _ptr(g2), _ptr(be2), float(eps2), _ptr(out), 0 if out is None else out.stride(0), int(out_pre_ln),
             None if out_planes is None else _ptr(out_planes.hi), None if out_planes is None else _ptr(out_planes.lo),
             0 if out_planes is None else out_planes.Cp, M, tg.K, tg.N, float(cscale), float(rscale),
             waves or tok_waves(M, (tg.K + 31) // 32), _stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * tg.K * tg.N, (M, tg.N, tg.K, 1, 3),
                     4.0 * (M * tg.K * (xdirs + (z is not None)) + tg.K * tg.N + M * tg.N * (1 + (res is not None)))))
    if out_planes is not None:
        return (out, out_planes) if out is not None else out_planes
    return out


# ---------------------------------------------------------------------------------------------- row kernels
def _mat(t: torch.Tensor):
    """(tensor, M, C, ld) for a map or a 2-D matrix."""
    if t.dim() == 2:
        assert t.stride(1) == 1
        return t, t.shape[0], t.shape[1], (t.stride(0) if t.stride(0) >= t.shape[1] else t.shape[1])
    return t, rows(t), t.shape[3], ld(t)


def _like(t: torch.Tensor, C: Optional[int] = None) -> torch.Tensor:
    if t.dim() == 2:
        C = C or t.shape[1]
        Cp = pad4(C)
        buf = zeros(t.shape[0], Cp, device=t.device) if Cp != C else torch.empty(t.shape[0], C, device=t.device)
        return buf[:, :C]
    return new_map(t.shape[0], t.shape[1], t.shape[2], C or t.shape[3], t.device)


def layernorm(x, gamma, beta, eps=1e-5, out=None, res1=None, res2=None, out_planes=None, want_f32=True, res2_vec=None,
              rows_per_batch=0):
    """out = LN(x) * gamma + beta (+ res1) (+ res2 [* res2_vec[batch, :]]).  out_planes: True / a Planes -> also emit the result as bf16 hi / lo
    planes for the planes GEMM (returns (out, planes), or only the planes when want_f32 is False)."""
    _, M, C, ldx = _mat(x)
    if out_planes is True:
        out_planes = Planes(1, 1, M, C, x.device) if x.dim() == 2 else Planes(x.shape[0], x.shape[1], x.shape[2], C, x.device)
    if out is None and (want_f32 or out_planes is None):
        out = _like(x)
    r1 = _mat(res1) if res1 is not None else (None, 0, 0, 0)
    r2 = _mat(res2) if res2 is not None else (None, 0, 0, 0)
    if res2_vec is not None:
        assert res2 is not None and res2_vec.is_contiguous() and res2_vec.shape[-1] == C and rows_per_batch > 0
    if out_planes is None and res2_vec is None:
        hip.call("ffsr_layernorm_f32", _ptr(x), ldx, _ptr(gamma), _ptr(beta), float(eps), _ptr(out), _mat(out)[3],
                 _ptr(r1[0]), r1[3], _ptr(r2[0]), r2[3], M, C, _stream())
        return out
    assert out_planes is None or (out_planes.M == M and out_planes.C == C)
    hip.call("ffsr_layernorm_planes_f32", _ptr(x), ldx, _ptr(gamma), _ptr(beta), float(eps), _ptr(out),
             0 if out is None else _mat(out)[3], None if out_planes is None else _ptr(out_planes.hi),
             None if out_planes is None else _ptr(out_planes.lo), 0 if out_planes is None else out_planes.Cp, _ptr(r1[0]), r1[3],
             _ptr(r2[0]), r2[3], _ptr(res2_vec), rows_per_batch, M, C, _stream())
    if out_planes is None:
        return out
    return (out, out_planes) if out is not None else out_planes


def final_norm(t2d, norm, B, H, W):
    """Last LayerNorm of a Swin-family body -> the input of conv_after_body: planes (when enabled) or an fp32 map."""
    if PLANES_AUTO and GEMM_MODE == "bf16x3":
        return layernorm(t2d, *norm, out_planes=True, want_f32=False).reshape_map(B, H, W)
    t = layernorm(t2d, *norm)
    s = t.stride(0)
    return t.as_strided((B, H, W, t.shape[1]), (H * W * s, W * s, s, 1), t.storage_offset())


def unary(x, act=ACT_NONE, slope=0.0, pre=1.0, alpha=1.0, beta=0.0, cscale=None, cbias=None, clamp=None, out=None):
    _, M, C, ldx = _mat(x)
    if out is None:
        out = _like(x)
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    hip.call("ffsr_unary_f32", _ptr(x), ldx, _ptr(out), _mat(out)[3], M, C, act, float(slope), float(pre),
             float(alpha), float(beta), _ptr(cscale), _ptr(cbias), int(clamp is not None), float(lo), float(hi), _stream())
    return out


def scale_add(a, b=None, *, avec=None, bvec=None, rows_per_batch=None, alpha=1.0, beta=1.0, out=None):
    """out = alpha * a * avec[n] + beta * b * bvec[batch, n]"""
    _, M, C, lda = _mat(a)
    if out is None:
        out = _like(a)
    ldb = _mat(b)[3] if b is not None else 0
    if rows_per_batch is None:
        rows_per_batch = M if a.dim() == 2 else a.shape[1] * a.shape[2]
    hip.call("ffsr_scale_add_f32", _ptr(a), lda, _ptr(avec), _ptr(b), ldb, _ptr(bvec), rows_per_batch, _ptr(out),
             _mat(out)[3], M, C, float(alpha), float(beta), _stream())
    return out


def mul_add(a, b, *, row_broadcast=False, c=None, alpha=1.0, gamma=1.0, out=None):
    """out = alpha * a * b (+ gamma * c); row_broadcast: b has one channel per pixel."""
    _, M, C, lda = _mat(a)
    if out is None:
        out = _like(a)
    ldc = _mat(c)[3] if c is not None else 0
    hip.call("ffsr_mul_add_f32", _ptr(a), lda, _ptr(b), _mat(b)[3], int(row_broadcast), _ptr(c), ldc, _ptr(out),
             _mat(out)[3], M, C, float(alpha), float(gamma), _stream())
    return out


def _nchunk(R: int) -> int:
    return max(1, min(2048, R // 64))


def _nchunk_gate(R: int) -> int:
    return max(1, min(4096, R // 128))


def colmean(x: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] -> [B,C] spatial mean."""
    B, H, W, C = x.shape
    n = _nchunk(H * W)
    part = torch.empty(B, n, C, device=x.device)
    out = torch.empty(B, C, device=x.device)
    hip.call("ffsr_colmean_f32", _ptr(x), ld(x), _ptr(out), _ptr(part), B, H * W, C, n, _stream())
    return out


def channel_attention(x: torch.Tensor, a1: Conv, a3: Conv) -> torch.Tensor:
    """[B,H,W,C] -> [B,C] = sigmoid(a3(relu(a1(spatial mean)))): the squeeze-excite tail of CAB, 2 launches."""
    B, H, W, C = x.shape
    n = max(1, min(256, H * W // 64))
    part = torch.empty(B, n, C, device=x.device)
    out = torch.empty(B, C, device=x.device)
    hip.call("ffsr_channel_attention_f32", _ptr(x), ld(x), _ptr(part), n, _ptr(a1.wgt), a1.wgt.shape[1], _ptr(a1.bias),
             _ptr(a3.wgt), a3.wgt.shape[1], _ptr(a3.bias), _ptr(out), B, H * W, C, a1.N, _stream())
    return out


@dataclass
class DwConv:
    w: torch.Tensor           # [KH*KW, C]
    bias: Optional[torch.Tensor]
    KH: int
    KW: int
    ph: int
    pw: int


def pack_dwconv(w: torch.Tensor, b, device, pad=None) -> DwConv:
    C, one, KH, KW = w.shape
    assert one == 1
    ph, pw = (KH // 2, KW // 2) if pad is None else pad
    return DwConv(w.detach().float().reshape(C, KH * KW).t().contiguous().to(device),
                  None if b is None else b.detach().float().contiguous().to(device), KH, KW, ph, pw)


def dwconv2d(x, dw: DwConv, act=ACT_NONE, out=None):
    B, H, W, C = x.shape
    if out is None:
        out = _like(x)
    hip.call("ffsr_dwconv2d_f32", _ptr(x), ld(x), _ptr(dw.w), _ptr(dw.bias), _ptr(out), ld(out), B, H, W, C, dw.KH,
             dw.KW, dw.ph, dw.pw, act, _stream())
    return out


def dw3x3_gate_pool(x, dw: DwConv):
    """x [B,H,W,2C] -> (gated [B,H,W,C], pooled mean [B,C])"""
    B, H, W, C2 = x.shape
    C = C2 // 2
    out = new_map(B, H, W, C, x.device)
    n = _nchunk_gate(H * W)
    part = torch.empty(B, n, C, device=x.device)
    pooled = torch.empty(B, C, device=x.device)
    hip.call("ffsr_dw3x3_gate_pool_f32", _ptr(x), ld(x), _ptr(dw.w), _ptr(dw.bias), _ptr(out), ld(out), _ptr(pooled),
             _ptr(part), B, H, W, C, n, _stream())
    return out, pooled


def bilinear(x, Ho, Wo, mul=1.0, out=None, accumulate=False):
    B, Hi, Wi, C = x.shape
    if out is None:
        out = new_map(B, Ho, Wo, C, x.device)
    hip.call("ffsr_bilinear_f32", _ptr(x), ld(x), _ptr(out), ld(out), B, Hi, Wi, Ho, Wo, C, float(mul),
             int(accumulate), _stream())
    return out


def bicubic_up(x, scale=4, out=None):
    B, H, W, C = x.shape
    if out is None:
        out = new_map(B, H * scale, W * scale, C, x.device)
    hip.call("ffsr_bicubic_up_f32", _ptr(x), ld(x), _ptr(out), ld(out), B, H, W, C, scale, _stream())
    return out


def avgpool2(x, out=None):
    B, H, W, C = x.shape
    if out is None:
        out = new_map(B, H // 2, W // 2, C, x.device)
    hip.call("ffsr_avgpool2_f32", _ptr(x), ld(x), _ptr(out), ld(out), B, H, W, C, _stream())
    return out


# ---------------------------------------------------------------------------------------------- attention / scan
def window_attn(qkv, bias, B, H, W, C, heads, ws, shift, scale, out=None, variant=None):
    """qkv [B*H*W, 3C] -> [B*H*W, C].  variant None: split-bf16 MFMA kernel in the default GEMM mode, the exact f32-MFMA
    kernel in FFSR_GEMM_MODE=f32; 0 / 3 force one of them (2: the exact kernel with 256 queries per workgroup)."""
    if out is None:
        out = torch.empty(B * H * W, C, device=qkv.device)
    if variant is None:
        variant = 0 if GEMM_MODE == "bf16x3" else 3
    hip.call("ffsr_window_attn_f32", _ptr(qkv), _mat(qkv)[3], _ptr(bias), _ptr(out), _mat(out)[3], B, H, W, C, heads,
             ws, shift, float(scale), variant, _stream())
    return out


def grl_window_attn(qkv, col0, biasT, logit, out, ocol0, B, H, W, heads, hd, shift):
    hip.call("ffsr_grl_window_attn_f32", _ptr(qkv), _mat(qkv)[3], col0, _ptr(biasT), _ptr(logit), _ptr(out),
             _mat(out)[3], ocol0, B, H, W, heads, hd, shift, _stream())
    return out


def grl_stripe_attn(qkv, col0, anchor, bias1T, bias2T, logit1, logit2, out, ocol0, B, H, W, heads, hd):
    hip.call("ffsr_grl_stripe_attn_f32", _ptr(qkv), _mat(qkv)[3], col0, _ptr(anchor), ld(anchor), _ptr(bias1T),
             _ptr(bias2T), _ptr(logit1), _ptr(logit2), _ptr(out), _mat(out)[3], ocol0, B, H, W, heads, hd, _stream())
    return out


def pixel_mha(qkv, S, T, E, heads, out=None, p_drop=0.0, seed=0):
    """p_drop > 0 (training): attention dropout with the counter-based mask of draw `seed` (see include/ffsr.h)"""
    if out is None:
        out = torch.empty(S * T, E, device=qkv.device)
    hip.call("ffsr_pixel_mha_f32", _ptr(qkv), _mat(qkv)[3], _ptr(out), _mat(out)[3], S, T, E, heads, float(p_drop), int(seed),
             _stream())
    return out


def scan_chunk(L: int, waves_per_chunk: int = 24) -> int:
    """Chunk length of the 3-pass scan.  One wave = (64 channels, one direction, one chunk); the kernels hold 4 waves per
    SIMD = 4096 resident waves on MI355X.  Long sequences: pick the chunk length (192..512 steps) whose wave count fills
    whole rounds of 4096 (measured at L = 180224: 352-step chunks = exactly 3 rounds, 17 % faster than 256-step chunks =
    4.1 rounds).  Short sequences: ~512 chunks, at least 32 steps each."""
    if L >= 64 * 1024:
        best = None
        for c in range(192, 513, 8):
            n = (L + c - 1) // c
            cost = -(-(n * waves_per_chunk) // 4096) * c
            if best is None or cost < best[0]:
                best = (cost, c)
        return best[1]
    c = 32
    while c * 2 <= min(1024, L // 512):
        c *= 2
    return c


# FFSR_SCAN_PAIRS=1: the scan folds its four per-direction outputs into two planes (directions 2 / 3 add to what 0 / 1 wrote).
# Off by default: measured at 352 x 512 x 360 the consumer gains what the scan loses (norm_gate 330 -> 212 us, scan 1756 -> 1877 us:
# two half-size launches per pass quantise worse over the 4096 resident waves, and the read-modify-write is not free)
SCAN_PAIRS = os.environ.get("FFSR_SCAN_PAIRS", "0") == "1"


def selective_scan4(u, xdbl, dtw, dtb, A, Dv, B, H, W, Dm, R, chunk=None, pairs=None):
    """u [B*L, Dm], xdbl [B*L, 4*(R+32)] -> y [4, B*L, Dm] (per-direction outputs in pixel order), or with `pairs` y [2, B*L, Dm] =
    (y0 + y2, y1 + y3): the second pair of directions is scanned after the first and adds to its planes."""
    L = H * W
    pairs = (SCAN_PAIRS and L >= 16384) if pairs is None else pairs
    nk = 2 if pairs else 4
    chunk = chunk or scan_chunk(L, nk * B * ((Dm + 63) // 64))
    nchunk = (L + chunk - 1) // chunk
    y = torch.empty(nk, B * L, Dm, device=u.device)
    hstate = torch.empty(B, nk, nchunk, Dm, 16, device=u.device)
    decay = torch.empty_like(hstate)
    hip.call("ffsr_selective_scan4_pairs_f32" if pairs else "ffsr_selective_scan4_f32", _ptr(u), _mat(u)[3], _ptr(xdbl), _mat(xdbl)[3], _ptr(dtw), _ptr(dtb), _ptr(A),
             _ptr(Dv), _ptr(y), Dm, _ptr(hstate), _ptr(decay), B, H, W, Dm, R, 16, chunk, _stream())
    return y


def mamba_norm_gate(y4, z, gamma, beta, eps=1e-5, out=None, out_planes=None, want_f32=True):
    """y4 [4, M, C] (or the [2, M, C] pair planes of selective_scan4), z [M, C] (row stride ldz) -> LayerNorm(sum_k y4[k]) * silu(z);
    out_planes as in layernorm()."""
    nd, M, C = y4.shape
    assert nd in (2, 4)
    if out_planes is True:
        out_planes = Planes(1, 1, M, C, y4.device)
    if out is None and (want_f32 or out_planes is None):
        out = torch.empty(M, C, device=y4.device)
    hip.call("ffsr_mamba_norm_gate_planes_f32" if nd == 4 else "ffsr_mamba_norm_gate_pairs_f32", _ptr(y4), M * C, C, _ptr(z), _mat(z)[3], _ptr(gamma), _ptr(beta), float(eps),
             _ptr(out), 0 if out is None else _mat(out)[3], None if out_planes is None else _ptr(out_planes.hi),
             None if out_planes is None else _ptr(out_planes.lo), 0 if out_planes is None else out_planes.Cp, M, C, _stream())
    if out_planes is not None:
        return (out, out_planes) if out is not None else out_planes
    return out


# ---------------------------------------------------------------------------------------------- fusion specials
def dct_bands(img, D, masks, scale, bands):
    B, H, W, _ = img.shape
    hip.call("ffsr_dct_bands_f32", _ptr(img), ld(img), _ptr(D), _ptr(masks), _ptr(scale), _ptr(bands), 36, B, H, W,
             _stream())


def dwt_db4(img, lo, hi):
    B, H, W, _ = img.shape
    Hd, Wd = (H + 6) // 2 + 1, (W + 6) // 2 + 1
    sub = zeros(B, Hd, Wd, 16, device=img.device)
    hip.call("ffsr_dwt_db4_f32", _ptr(img), ld(img), _ptr(lo), _ptr(hi), _ptr(sub), B, H, W, _stream())
    return sub


def fft_bands(img, twW, twH, mask, scale, bands):
    B, H, W, _ = img.shape
    work = torch.empty(10 * B * 3 * H * (W // 2 + 1), device=img.device)
    hip.call("ffsr_fft_bands_f32", _ptr(img), ld(img), _ptr(twW), _ptr(twH), _ptr(mask), _ptr(scale), _ptr(work),
             _ptr(bands), 36, B, H, W, _stream())


def selector_gates(raw, diff, temperature):
    B, h, w, _ = raw.shape
    gates = torch.empty(B, h, w, 4, device=raw.device)
    hip.call("ffsr_selector_gates_f32", _ptr(raw), ld(raw), _ptr(diff), ld(diff), _ptr(temperature), _ptr(gates), 4,
             B * h * w, _stream())
    return gates


def modulate(t_lr, w2, b2, img, out):
    B, h, w, _ = t_lr.shape
    _, Hh, Wh, _ = img.shape
    hip.call("ffsr_modulate_f32", _ptr(t_lr), ld(t_lr), _ptr(w2), _ptr(b2), _ptr(img), ld(img), _ptr(out), ld(out), B,
             h, w, Hh, Wh, _stream())


def fusion_route(enh, hier, routing, fw, gates, diff, out):
    B, Hh, Wh, _ = enh.shape
    _, h, w, _ = routing.shape
    hip.call("ffsr_fusion_route_f32", _ptr(enh), ld(enh), _ptr(hier), ld(hier), _ptr(routing), ld(routing), _ptr(fw),
             _ptr(gates), 0 if gates is None else ld(gates), _ptr(diff), 0 if diff is None else ld(diff), _ptr(out), ld(out),
             B, h, w, Hh, Wh, _stream())


def edge_final(sr, edge, gate, strength, lr, rscale, out):
    B, Hh, Wh, _ = sr.shape
    _, h, w, _ = lr.shape
    hip.call("ffsr_edge_final_f32", _ptr(sr), ld(sr), _ptr(edge), 0 if edge is None else ld(edge), _ptr(gate),
             0 if gate is None else ld(gate), _ptr(strength), _ptr(lr), ld(lr), _ptr(rscale), _ptr(out), ld(out), B, h, w, Hh, Wh,
             _stream())


# ---------------------------------------------------------------------------------------------- image boundary
def u8_to_map(img_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [B,H,W,3] (device) -> float map [B,H,W,3] / 255 with ld 4 (pad channel 0)."""
    B, H, W, C = img_u8.shape
    assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.is_cuda
    out = new_map(B, H, W, C, img_u8.device)
    hip.call("ffsr_u8_to_f32", _ptr(img_u8), _ptr(out), ld(out), B * H * W, C, _stream())
    return out


def map_to_u8(x: torch.Tensor) -> torch.Tensor:
    B, H, W, C = x.shape
    out = torch.empty(B, H, W, C, dtype=torch.uint8, device=x.device)
    hip.call("ffsr_f32_to_u8", _ptr(x), ld(x), _ptr(out), B * H * W, C, _stream())
    return out


def pad_reflect(x, Hp, Wp):
    B, H, W, C = x.shape
    if (Hp, Wp) == (H, W):
        return x
    out = new_map(B, Hp, Wp, C, x.device)
    hip.call("ffsr_pad_reflect_f32", _ptr(x), ld(x), _ptr(out), ld(out), B, H, W, Hp, Wp, C, _stream())
    return out


def crop(x, Ho, Wo, clamp=False, out=None, y0=0, x0=0):
    """window [y0:y0+Ho, x0:x0+Wo] of every image (compacting copy), optionally clamped to [0, 1]"""
    B, H, W, C = x.shape
    assert y0 + Ho <= H and x0 + Wo <= W
    if out is None:
        if (Ho, Wo) == (H, W) and not clamp:
            return x
        out = new_map(B, Ho, Wo, C, x.device)
    s = ld(x)
    hip.call("ffsr_crop_f32", x.data_ptr() + 4 * (y0 * W + x0) * s, s, _ptr(out), ld(out), B, H, W, Ho, Wo, C,
             int(clamp), _stream())
    return out


# ---------------------------------------------------------------------------------------------- geometric TTA
def _rot_map(k, h, w):
    """torch.rot90(x, k, [H, W]) as out[i, j] = x[f(i, j)] for an input of size h x w -> (f, (out_h, out_w))"""
    k %= 4
    if k == 0:
        return (lambda i, j: (i, j)), (h, w)
    if k == 1:
        return (lambda i, j: (j, w - 1 - i)), (w, h)
    if k == 2:
        return (lambda i, j: (h - 1 - i, w - 1 - j)), (h, w)
    return (lambda i, j: (h - 1 - j, i)), (w, h)


def dihedral(x, hflip: bool, rot: int, inverse=False, out=None, scale=1.0, accumulate=False):
    """forward: hflip (flip W) then rot90 x rot (scripts/extract_test_tta_cache.py:97-104);
    inverse: rot90 x (-rot) then hflip (scripts/generate_fast_submission.py:55-61).  x [B,H,W,C] map."""
    B, H, W, C = x.shape
    if not inverse:
        rmap, (Ho, Wo) = _rot_map(rot, H, W)
        f = (lambda i, j: (lambda y, xx: (y, W - 1 - xx))(*rmap(i, j))) if hflip else rmap
    else:
        rmap, (Ho, Wo) = _rot_map(-rot, H, W)
        f = (lambda i, j: rmap(i, Wo - 1 - j)) if hflip else rmap
    (cy, cx), (y1, x1), (y2, x2) = f(0, 0), f(1, 0), f(0, 1)
    if out is None:
        out = new_map(B, Ho, Wo, C, x.device)
    assert tuple(out.shape) == (B, Ho, Wo, C)
    hip.call("ffsr_dihedral_f32", _ptr(x), ld(x), _ptr(out), ld(out), B, H, W, Ho, Wo, C, y1 - cy, y2 - cy, cy, x1 - cx,
             x2 - cx, cx, float(scale), int(accumulate), _stream())
    return out
