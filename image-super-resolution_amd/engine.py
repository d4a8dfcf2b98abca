"""Per-image pipeline of the NTIRE entry point on the HIP engine.

Host-side mirror of models/team29_FreqFusionSR/io.py: ``Engine.__init__`` = _load_all_models :126 (weights
resident on the device, packed once), ``Engine.process`` = _process_image :222 (pad16 -> DRCT -> GRL ->
NAFNet -> MambaIR -> crops / clamps -> fusion.forward_with_precomputed).  Differences that do not change
results: no per-image mask/table rebuilds or H2D copies, no empty_cache().
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

from . import ops
from .drct import DRCT
from .fusion import EXPERTS, FusionNet
from .grl import GRL
from .mambair import MambaIR
from .nafnet import NAFNetSR


def require_gpu(device) -> torch.device:
    device = torch.device(device if device is not None else "cuda")
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("the FreqFusionSR HIP engine needs an MI355X (ROCm) device; there is no CPU fallback")
    return device


class Engine:
    def __init__(self, weights: Dict[str, dict], device=None, scale=4, fusion_flags=None):
        """fusion_flags: model.fusion.improvements of configs/train_config.yaml (io.py:186-193); None = all enabled."""
        self.device = require_gpu(device)
        self.scale = scale
        # weight preparation (packing, BN folding, bias tables) runs as torch ops on whatever device the tensors already
        # live on: after shard.broadcast_weights that is THIS GPU (views of the one broadcast blob), so nothing is staged
        # back through the host; a state_dict read from disk is on the CPU and is uploaded tensor by tensor while packing
        weights = {m: {k: (v.detach().float() if v.is_floating_point() else v.detach()) for k, v in sd.items()}
                   for m, sd in weights.items()}
        with torch.cuda.device(self.device):
            self.drct = DRCT(weights["drct"], self.device)
            self.grl = GRL(weights["grl"], self.device)
            self.nafnet = NAFNetSR(weights["nafnet"], self.device, scale)
            self.mamba = MambaIR(weights["mamba"], self.device)
            self.fusion = FusionNet(weights["fusion"], self.device, scale, fusion_flags)
            self.concurrent_experts = os.environ.get("FFSR_CONCURRENT_EXPERTS", "1") != "0"
            # two "lanes" (a main stream + four expert streams each): consecutive images may be submitted to alternate
            # lanes so that the latency-bound kernels of one image fill the gaps of the other (process(..., lane=i % 2))
            # (stream priorities measured in round 2 -- MambaIR's, DRCT's or both streams high: 374-380 ms vs 367 with equal
            #  priorities -- so all four expert streams stay at the default priority)
            self._lanes = [(torch.cuda.Stream(self.device), [torch.cuda.Stream(self.device) for _ in range(5)])
                           for _ in range(2)]       # per lane: a main stream, four expert streams and one for the fusion's LR-only phases
            self._streams = self._lanes[0][1]
            self._graphs = {}

    # -------------------------------------------------------------------------------------- experts
    def run_experts(self, lr, streams=None, with_lr_phases=False):
        """lr [B,h,w,3] float map -> (imgs, feats) as io._process_image builds them (io.py:224-278).
        The four experts are independent: each runs on its own HIP stream so their launch tails and small kernels
        overlap; the caller's stream waits for all four before the fusion starts.  with_lr_phases: also run the fusion
        network's LR-only phases (frequency bands, cross-band routing, selector gates: dozens of small, latency-bound
        launches) on a fifth stream meanwhile -> (imgs, feats, pre)."""
        B, h, w, _ = lr.shape
        s = self.scale
        hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
        if with_lr_phases:
            # input-independent tables of this image size (DFT twiddles, FFT mask) are built on the CALLER's stream, before
            # the fork: every side stream (and the other lane) orders itself behind `ready` / this stream
            self.fusion.prepare(h, w)
        lp = ops.pad_reflect(lr, hp, wp)
        imgs, feats = {}, {}

        def swin_like(name, model):
            sr, f = model(lp)
            imgs[name] = ops.crop(sr, h * s, w * s, clamp=True)
            feats[name] = ops.crop(f, h, w)

        def naf():
            sr, f = self.nafnet(lp)
            imgs["nafnet"] = ops.crop(sr, h * s, w * s)
            feats["nafnet"] = ops.bilinear(f, h, w)          # padded HR map straight to (h, w): io.py:256-258

        jobs = [lambda: swin_like("drct", self.drct), lambda: swin_like("mamba", self.mamba),
                lambda: swin_like("grl", self.grl), naf]
        pre = []
        if with_lr_phases:
            jobs.append(lambda: pre.extend(self.fusion.lr_phases(lr)))
        if not self.concurrent_experts:
            for job in jobs:
                job()
            return (imgs, feats, tuple(pre)) if with_lr_phases else (imgs, feats)
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        streams = streams or self._streams
        for job, stream in zip(jobs, streams):
            stream.wait_event(ready)
            with torch.cuda.stream(stream):
                job()
        for stream in streams[:len(jobs)]:
            main.wait_stream(stream)
        for t in list(imgs.values()) + list(feats.values()) + [lp] + list(pre):
            t.record_stream(main)
        return (imgs, feats, tuple(pre)) if with_lr_phases else (imgs, feats)

    # tiles of at most this many LR pixels (batch 1) are launch-bound when enqueued eagerly (~3.5 k launches, ~40 ms of host
    # time per 64x64 tile against ~15 ms of GPU time): process() replays them from a captured HIP graph by default
    GRAPH_MAX_PIXELS = 128 * 128

    def process(self, lr, lane=None, graph=None):
        """lr [B,h,w,3] float map in [0,1] -> SR map [B,4h,4w,3] in [0,1].
        lane None: runs on the caller's current stream.  lane 0 / 1: runs asynchronously on that lane's own streams
        (the caller's stream is only waited for at the start); the caller must ``join()`` before reading the result.
        graph None: small single tiles (<= 128x128 LR pixels) are replayed from a HIP graph captured once per shape (the result
        is copied out of the graph's static buffer, so it stays valid across calls); True / False force it."""
        B, h, w, _ = lr.shape
        if graph is None:
            graph = (lane is None and B == 1 and h * w <= self.GRAPH_MAX_PIXELS and self.concurrent_experts
                     and os.environ.get("FFSR_GRAPH_SMALL", "1") != "0" and not torch.cuda.is_current_stream_capturing())
        if graph:
            return self.process_graphed(lr).clone()
        with torch.cuda.device(self.device):
            if lane is None:
                imgs, feats, pre = self.run_experts(lr, with_lr_phases=True)
                return self.fusion(lr, imgs, feats, pre=pre)
            main, streams = self._lanes[lane]
            main.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(main):
                lr.record_stream(main)
                imgs, feats, pre = self.run_experts(lr, streams, with_lr_phases=True)
                return self.fusion(lr, imgs, feats, pre=pre)

    def process_graphed(self, lr):
        """process() replayed from a HIP graph captured once per input shape (hipGraph via torch.cuda.CUDAGraph: the
        ~3.5 k launches of an image, incl. the fork / join of the four expert streams, become one graph launch).  Pays
        on small tiles, where the eager path is launch-bound (64x64: 39 -> 20 ms); at 340x510 the kernels dominate.
        The result lives in the graph's static output buffer: consume (or clone) it before the next call of that shape."""
        with torch.cuda.device(self.device):
            key = tuple(lr.shape)
            ent = self._graphs.get(key)
            if ent is None:
                static_in = lr.clone()
                side = torch.cuda.Stream(self.device)            # warm-up off the capture: first-launch attributes,
                side.wait_stream(torch.cuda.current_stream())    # lazily packed constants and allocator pools settle
                with torch.cuda.stream(side):
                    self.process(static_in, graph=False)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self.process(static_in, graph=False)
                ent = self._graphs[key] = (graph, static_in, static_out)
            graph, static_in, static_out = ent
            static_in.copy_(lr)
            graph.replay()
            return static_out

    def join(self):
        """The caller's current stream waits for both lanes."""
        cur = torch.cuda.current_stream(self.device)
        for main, _ in self._lanes:
            cur.wait_stream(main)

    def process_tta(self, lr):
        """8x geometric self-ensemble: mean over {hflip} x {rot90^k} of the de-transformed outputs,
        clamped (scripts/extract_test_tta_cache.py:253-256, scripts/generate_fast_submission.py:235-250)."""
        with torch.cuda.device(self.device):
            acc = None
            for hflip in (False, True):
                for rot in range(4):
                    sr = self.process(ops.dihedral(lr, hflip, rot))
                    acc = ops.dihedral(sr, hflip, rot, inverse=True, out=acc, scale=0.125, accumulate=acc is not None)
            return ops.unary(acc, clamp=(0.0, 1.0), out=acc)

    # -------------------------------------------------------------------------------------- uint8 boundary
    def upload(self, img_u8: np.ndarray) -> torch.Tensor:
        """uint8 HxWx3 RGB (host) -> float map [1,h,w,3] on the device (io._uint2tensor4)."""
        t = torch.from_numpy(np.ascontiguousarray(img_u8)).to(self.device, non_blocking=True)
        return ops.u8_to_map(t.unsqueeze(0))

    def download(self, sr) -> np.ndarray:
        """SR map [1,H,W,3] -> uint8 HxWx3 (io._tensor2uint: clamp, *255, round half to even)."""
        return ops.map_to_u8(sr)[0].cpu().numpy()

    def process_u8(self, img_u8: np.ndarray) -> np.ndarray:
        with torch.cuda.device(self.device):
            return self.download(self.process(self.upload(img_u8)))


# ---------------------------------------------------------------------------------------------- layout helpers
def nchw_to_map(x: torch.Tensor, device) -> torch.Tensor:
    """[B,C,H,W] (any device) -> channels-last map on `device` with the stride padded to a multiple of 4."""
    B, C, H, W = x.shape
    m = ops.new_map(B, H, W, C, device)
    m.copy_(x.to(device).permute(0, 2, 3, 1))
    return m


def map_to_nchw(m: torch.Tensor) -> torch.Tensor:
    return m.permute(0, 3, 1, 2).contiguous().cpu()
