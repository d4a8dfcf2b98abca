"""The reference's cached-feature training step on the HIP kernels.

``FusionTrainer.step`` is the body of ``train_epoch_cached`` (train.py:297-359).  Around ``loss.backward()`` it works on ONE
flat fp32 parameter buffer (``FusionOptimizer``): ``sr.clamp(0, 1)`` + ``L1Loss`` + the division by ``accumulation_steps``
(:326-336), ``clip_grad_norm_`` (:347-352), ``optimizer.step()`` of ``torch.optim.AdamW`` (:354) and ``EMAModel.update``
(:358-359, checkpoint_manager.py:349-356) -- kernels of ffsr_train.hip.  The train-mode forward and the backward pass
themselves are fusion_train.FusionTrainNet on autograd.Tape (kernels of the forward library + ffsr_backward.hip /
ffsr_wgrad.hip); they fill ``FusionOptimizer.grad``.  No CPU fallback: the calls raise without the HIP library.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import hip
from .ops import _mat, _ptr, _stream

N_PARTIAL = 1024


def l1_clamp_loss(sr: torch.Tensor, hr: torch.Tensor, accumulation_steps: int = 1, need_grad: bool = True):
    """sr, hr: channels-last maps [B,H,W,C] (row strides may be padded).  Returns (loss [1] on the device =
    mean|clamp(sr,0,1) - hr| / accumulation_steps, d loss / d sr with sr's shape, or None)."""
    _, M, C, lds = _mat(sr)
    _, M2, C2, ldh = _mat(hr)
    if (M, C) != (M2, C2):
        raise ValueError(f"sr {tuple(sr.shape)} and hr {tuple(hr.shape)} differ")
    grad = torch.empty_like(sr) if need_grad else None
    part = torch.empty(N_PARTIAL, device=sr.device)
    loss = torch.empty(1, device=sr.device)
    hip.call("ffsr_l1_clamp_loss_f32", _ptr(sr), lds, _ptr(hr), ldh, _ptr(grad), 0 if grad is None else _mat(grad)[3],
             _ptr(part), N_PARTIAL, _ptr(loss), M, C, 1.0 / accumulation_steps, _stream())
    return loss, grad


class FusionOptimizer:
    """AdamW + gradient clipping + EMA over the trainable tensors of a state_dict, held in one flat device buffer.
    Defaults = configs/train_config.yaml of the reference (lr 2e-4, weight decay 1e-4, betas (0.9, 0.999), eps 1e-8,
    gradient_clip 1.0, EMA decay 0.999)."""

    def __init__(self, params: Dict[str, torch.Tensor], device, lr=2e-4, betas: Tuple[float, float] = (0.9, 0.999), eps=1e-8,
                 weight_decay=1e-4, max_norm=1.0, ema_decay=0.999):
        self.device = torch.device(device)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_norm, self.ema_decay = max_norm, ema_decay
        self.layout = []
        off = 0
        for k, v in params.items():
            self.layout.append((k, tuple(v.shape), off, v.numel()))
            off += v.numel()
        self.n = off
        self.param = torch.empty(off, device=self.device)
        for (k, shape, o, n) in self.layout:
            self.param[o:o + n] = params[k].detach().reshape(-1).to(self.device, torch.float32)
        self.grad = torch.zeros_like(self.param)
        self.exp_avg = torch.zeros_like(self.param)
        self.exp_avg_sq = torch.zeros_like(self.param)
        self.ema = self.param.clone() if ema_decay is not None else None       # EMAModel.__init__: shadow = clone
        self.step_count = 0
        self._part = torch.empty(N_PARTIAL, device=self.device)
        self._sumsq = torch.zeros(1, device=self.device)

    def views(self, buf: torch.Tensor = None) -> Dict[str, torch.Tensor]:
        """name -> view into the flat buffer (default: the parameters)"""
        buf = self.param if buf is None else buf
        return {k: buf[o:o + n].reshape(shape) for (k, shape, o, n) in self.layout}

    def zero_grad(self):
        self.grad.zero_()

    def grad_norm(self) -> torch.Tensor:
        """total 2-norm of the gradient (device scalar), as clip_grad_norm_ returns it"""
        hip.call("ffsr_sumsq_f32", _ptr(self.grad), self.n, _ptr(self._part), N_PARTIAL, _ptr(self._sumsq), _stream())
        return self._sumsq.sqrt()

    def step(self, lr: float = None):
        """clip_grad_norm_ -> AdamW.step -> EMA.update, two launches + one fused pass, no host synchronisation"""
        self.step_count += 1
        clip = self.max_norm is not None and self.max_norm > 0
        if clip:
            hip.call("ffsr_sumsq_f32", _ptr(self.grad), self.n, _ptr(self._part), N_PARTIAL, _ptr(self._sumsq), _stream())
        hip.call("ffsr_adamw_ema_f32", _ptr(self.param), _ptr(self.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                 _ptr(self.ema), self.n, _ptr(self._sumsq) if clip else None, float(self.max_norm or 0.0),
                 float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                 float(self.weight_decay), self.step_count, float(self.ema_decay or 0.0), _stream())


class FusionTrainer:
    """One cached-feature training step of the reference (train.py:297-359, ``train_epoch_cached``) on the HIP kernels:

        sr = model.forward_with_precomputed(lr, expert_imgs, expert_feats)     # model.train(): fusion_train.FusionTrainNet
        loss = L1Loss()(sr.clamp(0, 1), hr) / accumulation_steps               # ffsr_l1_clamp_loss_f32
        loss.backward()                                                        # autograd.Tape.backward()
        clip_grad_norm_ / optimizer.step() / zero_grad / ema.update            # FusionOptimizer.step (every accumulation_steps)

    ``state_dict``: the fusion network's (parameters AND buffers; reference keys).  attn_dropout: p of the two
    nn.MultiheadAttention's attention dropout (default 0.1 as the reference; 0 = the fixture-pinned configuration), seed: of
    its counter-based masks (see fusion_train.py).  No CPU fallback."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device, scale=4, accumulation_steps=1, attn_dropout=0.1, seed=0,
                 **opt_kwargs):
        from . import autograd, fusion_train
        self.device = torch.device(device)
        self.accumulation_steps = accumulation_steps
        sd = {k: v for k, v in state_dict.items() if v.is_floating_point() and v.numel() > 0}
        # BatchNorm's int64 update counters (six ``*.num_batches_tracked`` buffers): kept on the host, advanced by the
        # number of train-mode applications of their layer (BnP.calls) and written back by state_dict(), so that the key
        # set equals CompleteEnhancedFusionSR.state_dict()'s and a resumed run continues the count
        self._nbt0 = {k: int(v) for k, v in state_dict.items() if k.endswith(".num_batches_tracked")}
        params = {k: v for k, v in sd.items() if fusion_train.is_parameter(k)}
        with torch.cuda.device(self.device):
            self.opt = FusionOptimizer(params, self.device, **opt_kwargs)
            self.buffers = {k: v.detach().to(self.device, torch.float32).contiguous().clone() for k, v in sd.items()
                            if not fusion_train.is_parameter(k)}
            pv, gv = self.opt.views(), self.opt.views(self.opt.grad)
            self.params = {k: autograd.Param(k, pv[k], gv[k]) for k in pv}
            self.net = fusion_train.FusionTrainNet(self.params, self.buffers, self.device, scale, attn_dropout=attn_dropout)
            self.tape = autograd.Tape(self.device)
            self.tape.seed = int(seed)
        self.forward_count = 0
        self.micro = 0

    def zero_grad(self):
        hip.call("ffsr_zero_f32", _ptr(self.opt.grad), self.opt.n, _stream())

    def forward_backward(self, lr, hr, imgs, feats):
        """lr [B,h,w,3], hr [B,4h,4w,3], imgs / feats: dicts of channels-last maps on the device (cached expert outputs).
        Accumulates d loss / d parameters into the flat gradient buffer; returns (loss [1] device tensor, sr map)."""
        with torch.cuda.device(self.device), torch.no_grad():
            self.tape.clear()      # closures left behind by a forward / backward that raised must never be replayed
            self.tape.new_step(self.forward_count)          # this pass's dropout draws
            self.forward_count += 1
            try:
                sr = self.net.forward(self.tape, lr, imgs, feats)
                loss, g = l1_clamp_loss(sr.v, hr, self.accumulation_steps)
                sr.g, sr.gown = g, True
                self.tape.backward()
            finally:
                self.tape.clear()
        return loss, sr.v

    def step(self, lr, hr, imgs, feats, lr_rate: float = None):
        """one micro-batch; the optimiser steps every `accumulation_steps` calls (train.py:344-359)"""
        if self.micro == 0:
            self.zero_grad()
        loss, sr = self.forward_backward(lr, hr, imgs, feats)
        self.micro += 1
        if self.micro == self.accumulation_steps:
            with torch.cuda.device(self.device):
                self.opt.step(lr_rate)
                self.net.repack()
            self.micro = 0
        return loss

    def state_dict(self, ema=False) -> Dict[str, torch.Tensor]:
        """reference-keyed state (parameters from the live weights or the EMA shadow, buffers incl. running statistics)"""
        out = {k: v.clone() for k, v in self.opt.views(self.opt.ema if ema else None).items()}
        out.update({k: v.clone() for k, v in self.buffers.items()})
        for q, bn in self.net.batchnorms.items():
            k = q + ".num_batches_tracked"
            out[k] = torch.tensor(self._nbt0.get(k, 0) + bn.calls, dtype=torch.long)
        return out
