#!/usr/bin/env python3
"""Benchmark of the FreqFusionSR x4 hot path on MI355X (contract: see the task description / DESIGN.md).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one 340x510 LR image (BASELINE.json's metric size: 510x340 -> 2040x1360) through the full hot path --
pad16 -> DRCT-L, GRL-B, NAFNet-w64, MambaIR -> 7-phase fusion -- with random-init weights of the exact architecture
and a synthetic LR image already resident in HBM.  value = output megapixels / s of the whole job (all ranks);
multi-GPU is image-parallel (weak scaling: every rank processes K images, weights broadcast once over RCCL, no
collective in the forward pass).

Also on the JSON line:
  roofline     -- the dominant kernel conv_gemm_kernel (f32 MFMA implicit GEMM: every linear / conv of the path):
                  algorithmic FLOPs per launch / mean launch duration, timed with events on the launch stream
                  during one extra instrumented (untimed-for-`value`) step.
  cpu_baseline -- the CPU oracle (a port of the reference's PyTorch path) on the box's host cores, rank 0 / N=1
                  only, on a bounded sample (one 64x64 LR tile, full-size experts).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_LR, W_LR, SCALE = 340, 510, 4
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s measured achievable)


def synth_lr(seed, h, w, b=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(b, 3, h, w, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), 5, 1)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return torch.floor(x * 256).clamp(0, 255) / 255.0


def cpu_baseline(weights, naf_cfg=None, tiles=5, full=False):
    """Oracle (torch-CPU port of the reference path) on 64x64 tiles, one untimed warm-up tile then `tiles` timed ones
    (the reference loop is batch 1: io.py:330-345); returns the cpu_baseline object."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from ffsr_oracle import pipeline
    from ffsr_oracle.scan_c import selective_scan_c          # same recurrence as scan.py, in C + OpenMP
    # SURVEY 8d asks for "all physical host cores, count stated".  A GPU box exposes the whole host (os.cpu_count() = 128+) but
    # grants one GPU's share of it, 16 cores: more OpenMP threads than that only time-slice (the 340x510 oracle test ran 4x
    # slower with 128 threads than with 16).  So: min(16, visible cores), stated in the object.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    weights = {m: {k: v.detach().float().cpu() for k, v in sd.items()} for m, sd in weights.items()}
    mp, dt = 0.0, 0.0
    with torch.no_grad():
        for i in range(tiles + 1):
            lr = synth_lr(1234 + i, 64, 64)
            t0 = time.perf_counter()
            out = pipeline.process_image(weights, lr, naf_cfg=naf_cfg, scan_fn=selective_scan_c)
            if i > 0:
                dt += time.perf_counter() - t0
                mp += out.shape[-1] * out.shape[-2] / 1e6
    out = {"value": mp / dt, "unit": "output MP/s", "cores": cores, "kind": "port",
           "cores_visible": os.cpu_count(), "cores_granted": cores,
           "sample": f"{tiles} x 64x64 LR tiles -> 256x256 ({mp:.3f} MP), 4 experts + fusion, {dt:.1f} s of CPU work after one warm-up tile; "
                     f"{cores} threads = the box's CPU share for one GPU ({os.cpu_count()} logical cores visible)",
           # the tile sample flatters the CPU (cache-resident maps): one real 340x510 oracle pass on the same 16-core share
           # measured 263 s = 0.0105 MP/s (profiles/r03_cpu_baseline_at_metric_size.md; re-measure with --cpu-full)
           "at_metric_size": {"value": 0.0105, "unit": "output MP/s", "seconds_per_image": 263.4, "measured": "round 3, GPU box, "
                              "16 threads, python bench.py --cpu-full (one 510x340 image, no warm-up)", "live": False}}
    if full:
        lr = synth_lr(1234, H_LR, W_LR)
        with torch.no_grad():
            t0 = time.perf_counter()
            o = pipeline.process_image(weights, lr, naf_cfg=naf_cfg, scan_fn=selective_scan_c)
            dt_f = time.perf_counter() - t0
        out["at_metric_size"] = {"value": o.shape[-1] * o.shape[-2] / 1e6 / dt_f, "unit": "output MP/s", "seconds_per_image": dt_f,
                                 "measured": f"this run, {cores} threads, one {W_LR}x{H_LR} image, no warm-up", "live": True}
    return out


def train_measure(args, steps, warmup, cpu=True, quiet=False):
    """BASELINE config 5 (`--config train`): ONE cached-feature training step of the fusion network per "step" --
    forward_with_precomputed in train mode on a batch of 32 cached 64x64 LR patches (seeded expert images / features
    resident in HBM), L1 loss after clamp, backward through all 1.43 M parameters, clip_grad_norm_(1.0), AdamW(2e-4,
    wd 1e-4), EMA(0.999): train.py:297-359 of the reference.  Frozen experts are NOT run (their outputs are the cache,
    exactly as in the reference's cached loop); attention dropout is off.  Single GPU by definition."""
    W = importlib.import_module("image-super-resolution_amd.weights")
    E = importlib.import_module("image-super-resolution_amd.engine")
    T = importlib.import_module("image-super-resolution_amd.train")
    ops = importlib.import_module("image-super-resolution_amd.ops")
    if args.gemm:
        ops.set_gemm_mode(args.gemm)
    if args.gpus != 1:
        raise SystemExit("--config train is a single-GPU configuration (BASELINE config 5)")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    B, h, w = args.batch if args.batch > 1 else 32, 64, 64
    g = torch.Generator().manual_seed(4321)
    sd = {k: v for k, v in W.fusion_state_dict(seed=0).items() if v.is_floating_point() and v.numel() > 0}
    tr = T.FusionTrainer(sd, device)
    lr = synth_lr(99, h, w, B)
    bic = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    names = ("drct", "grl", "nafnet", "mamba")
    imgs = {n: E.nchw_to_map((bic + 0.03 * torch.randn(bic.shape, generator=g)).clamp(0, 1), device) for n in names}
    feats = {n: E.nchw_to_map(torch.randn(B, 64 if n == "nafnet" else 180, h, w, generator=g), device) for n in names}
    hr = E.nchw_to_map(torch.rand(B, 3, 4 * h, 4 * w, generator=g), device)
    lrm = E.nchw_to_map(lr, device)
    T0 = time.perf_counter()
    for i in range(warmup):
        tr.step(lrm, hr, imgs, feats)
        torch.cuda.synchronize(device)
        if not quiet:
            print(f"[bench +{time.perf_counter() - T0:6.1f}s] warm-up training step {i} done", file=sys.stderr, flush=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.step(lrm, hr, imgs, feats)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    # fusion forward 116.3 GFLOP per 64x64 tile (112.8 conv/GEMM + 3.5 MHA); backward = input + weight gradient
    tflop = 3 * B * (h * w / 4096.0) * 116.3e9 / 1e12
    peak = MFMA_F32_PEAK_TFLOPS if ops.GEMM_MODE == "f32" else MFMA_BF16_PEAK_TFLOPS / 3.0
    line = {"metric": f"cached-feature training steps/s (BASELINE config 5: {B} x {w}x{h} LR patches, fusion net fwd + bwd + AdamW)",
            "value": steps / dt, "unit": "steps/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if ops.GEMM_MODE == "f32" else "f32 (conv products as 3-term split-bf16 MFMA; weight gradients: split-bf16 MFMA for the wide 3x3 layers, exact fp32 elsewhere)",
            "data": "synthetic (seeded cached expert images / features / HR targets; random-init fusion weights)",
            "config": {"workload": f"train_epoch_cached step: forward_with_precomputed(train) + L1 + backward + clip + AdamW + EMA, "
                                   f"batch {B} of {w}x{h} LR patches -> {4 * w}x{4 * h}", "patches_per_step": B,
                       "patches_per_s": B * steps / dt, "attention_dropout": tr.net.attn_dropout},
            "gemm_mode": ops.GEMM_MODE, "final_loss": float(loss.item()),
            "roofline": {"bound": "mfma", "achieved": tflop / (dt / steps), "peak": peak, "unit": "TFLOP/s",
                         "frac": tflop / (dt / steps) / peak, "traffic": None,
                         "note": "whole step: algorithmic 3 x forward FLOPs (SURVEY 8d) / step time, not a single kernel"}}
    if cpu and not args.no_cpu_baseline:
        # the same step on the host: the oracle's train-mode forward differentiated by torch autograd + torch.optim.AdamW, on a
        # bounded sample (2 of the 32 patches, one warm-up + 2 timed steps); reported in patches/s next to config.patches_per_s
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from ffsr_oracle import fusion as ofusion
        FT = importlib.import_module("image-super-resolution_amd.fusion_train")
        cores = min(16, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        nb = 2
        prm = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if FT.is_parameter(k)}
        sdo = {k: prm.get(k, v.clone().float()) for k, v in sd.items()}
        opt = torch.optim.AdamW(list(prm.values()), lr=2e-4, weight_decay=1e-4)
        c_imgs = {n: E.map_to_nchw(v[:nb]) for n, v in imgs.items()}
        c_feats = {n: E.map_to_nchw(v[:nb]) for n, v in feats.items()}
        c_hr, c_lr = E.map_to_nchw(hr[:nb]), lr[:nb]
        cdt = 0.0
        for i in range(3):
            t1 = time.perf_counter()
            opt.zero_grad()
            out = ofusion.fusion_forward(sdo, c_lr, c_imgs, c_feats, train=True)
            torch.nn.functional.l1_loss(out.clamp(0, 1), c_hr).backward()
            torch.nn.utils.clip_grad_norm_(list(prm.values()), 1.0)
            opt.step()
            if i > 0:
                cdt += time.perf_counter() - t1
        line["cpu_baseline"] = {"value": 2 * nb / cdt, "unit": "patches/s", "cores": cores, "kind": "port",
                                "sample": f"2 timed steps on {nb} of the {B} patches after one warm-up step ({cdt:.1f} s of CPU work): "
                                          f"oracle train-mode forward + torch autograd + clip + torch.optim.AdamW"}
        line["gpu_over_cpu"] = line["config"]["patches_per_s"] / line["cpu_baseline"]["value"]
    del tr
    torch.cuda.empty_cache()
    return line


def train_bench(args):
    print(json.dumps(train_measure(args, args.steps, args.warmup)))


def dry_run(args):
    """The multi-rank plumbing of the bench without the engine: rendezvous, weight broadcast (rank 0's values must reach
    every rank), barrier-bracketed timed region, MAX over ranks, one JSON line from rank 0.  CPU / gloo; marked dry_run."""
    W = importlib.import_module("image-super-resolution_amd.weights")
    S = importlib.import_module("image-super-resolution_amd.shard")
    rank, world = S.rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    S.init_process_group(args.backend)
    weights = {"fusion": W.fusion_state_dict(seed=5)} if rank == 0 else None
    templ = {"fusion": {k: torch.empty(v.shape, device="meta") for k, v in W.fusion_state_dict(seed=5).items()}}
    weights = S.broadcast_weights(weights if rank == 0 else templ, torch.device("cpu"))
    check = float(sum(v.double().sum() for v in weights["fusion"].values() if v.is_floating_point()))
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))            # the slowest rank sets the time
    if world > 1:
        torch.distributed.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    sums = torch.tensor([check], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        gathered = [torch.zeros_like(sums) for _ in range(world)]
        torch.distributed.all_gather(gathered, sums)
        assert all(float(g) == check for g in gathered), "broadcast weights differ between ranks"
    if rank == 0:
        print(json.dumps({"metric": "dry run (launch / rendezvous / broadcast plumbing only)", "dry_run": True, "value": None,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * float(tmax) / max(args.steps, 1), "weights_checksum": check}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["infer", "train"], default="infer",
                    help="infer: BASELINE's headline metric (default); train: BASELINE config 5, one cached-feature training step")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=H_LR)
    ap.add_argument("--width", type=int, default=W_LR)
    ap.add_argument("--graph", action="store_true", help="replay each step from a captured HIP graph (pays on small tiles)")
    ap.add_argument("--lanes", type=int, default=1, help="stream lanes: images in flight per GPU (1 or 2)")
    ap.add_argument("--batch", type=int, default=1, help="images per step per GPU (BASELINE config 3 uses 16 x 64x64)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline: also time ONE real 340x510 oracle pass (~5 min of host time)")
    ap.add_argument("--no-extras", action="store_true", help="skip the tile64 / train_step measurements after the headline one")
    ap.add_argument("--small", action="store_true", help="reduced-depth experts (plumbing check only, not a valid bench)")
    ap.add_argument("--gemm", choices=["f32", "bf16x3", "bf16"], default=None,
                    help="GEMM arithmetic (default: the engine's default, bf16x3; bf16 = plain bf16 operands, a precision option)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / broadcast / timing plumbing only, no engine and no GPU (CPU tests; not a bench)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process becomes the launcher -- it starts N fresh rank processes (one
        # per GPU, the reference's scheme: scripts/kaggle_inference_fixed.py:385-397), relays rank 0's JSON line and exits
        # with their status.  It never touches a GPU itself and is never replaced by another program.
        S = importlib.import_module("image-super-resolution_amd.shard")
        def relay(rank, line, is_err):       # stdout carries rank 0's JSON line and nothing else (gloo chats on stdout)
            out = sys.stdout if (rank == 0 and not is_err and line.lstrip().startswith("{")) else sys.stderr
            out.write(line)
            out.flush()

        procs, threads, _ = S.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], relay=relay)
        sys.exit(S.join_ranks(procs, threads))
    if args.config == "train":
        return train_bench(args)
    if args.dry_run:
        return dry_run(args)

    W = importlib.import_module("image-super-resolution_amd.weights")
    E = importlib.import_module("image-super-resolution_amd.engine")
    S = importlib.import_module("image-super-resolution_amd.shard")
    ops = importlib.import_module("image-super-resolution_amd.ops")

    T0 = time.perf_counter()
    if args.gemm:
        ops.set_gemm_mode(args.gemm)
    rank, world = S.rank_world()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    if "FFSR_BENCH_DEVICE" in os.environ:        # rehearsal: several ranks on one card
        local = int(os.environ["FFSR_BENCH_DEVICE"])
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    S.init_process_group(args.backend)

    # only the broadcast source needs values; the other ranks hand in the key / shape template (meta tensors)
    weights = W.random_weights(seed=0, small=args.small, shapes_only=(rank != 0))
    weights = S.broadcast_weights(weights, device)          # RCCL over xGMI, rank 0's values win (no-op at N=1)
    eng = E.Engine(weights, device)
    h, w = args.height, args.width
    lrs = [E.nchw_to_map(synth_lr(1234 + rank * 1000 + i, h, w, args.batch), device) for i in range(max(1, min(args.steps, 4)))]

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"engine ready; warm-up x{args.warmup} on {h}x{w} LR")
    step_fn = eng.process_graphed if args.graph else eng.process
    for i in range(args.warmup):
        step_fn(lrs[i % len(lrs)])
        torch.cuda.synchronize(device)
        log(f"warm-up step {i} done")
    barrier()
    t0 = time.perf_counter()
    outs = []
    for i in range(args.steps):
        # consecutive steps go to alternate stream lanes (two images in flight); --lanes 1 serialises them
        outs.append(eng.process_graphed(lrs[i % len(lrs)]) if args.graph else
                    eng.process(lrs[i % len(lrs)], lane=(i % args.lanes) if args.lanes > 1 else None))
        if len(outs) > 2:
            outs.pop(0)
    eng.join()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    log(f"timed {args.steps} steps: {1e3 * dt / args.steps:.1f} ms/step")

    mp_per_image = (h * SCALE) * (w * SCALE) / 1e6
    value = world * args.steps * args.batch * mp_per_image / dt

    # ---- roofline of the dominant kernel: one extra instrumented step on the same stream
    # (experts run one after the other here: with the four expert streams overlapping, an event pair around one
    #  launch would also time its neighbours)
    concurrent, eng.concurrent_experts = eng.concurrent_experts, False
    ops.CONV_PROFILE = []
    eng.process(lrs[0])
    torch.cuda.synchronize(device)
    prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    dur_ms = [a.elapsed_time(b) for a, b, *_ in prof]
    flops = sum(t[2] for t in prof)
    algo_bytes = sum(t[4] for t in prof)
    conv_s = sum(dur_ms) / 1e3
    t1 = time.perf_counter()
    eng.process(lrs[0])
    torch.cuda.synchronize(device)
    step_s = time.perf_counter() - t1          # sequential-experts step, the denominator of kernel_share_of_step
    eng.concurrent_experts = concurrent
    achieved = flops / conv_s / 1e12
    # per-expert split of one step (diagnostic, stderr only)
    if rank == 0:
        by_shape = {}
        for (e0, e1, f, shape, nbytes), ms in zip(prof, dur_ms):
            t = by_shape.setdefault(shape, [0, 0.0, 0.0, 0.0])
            t[0] += 1; t[1] += ms; t[2] += f; t[3] += nbytes
        for shape, (n, ms, f, nb) in sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('FFSR_BENCH_SHAPES', '14'))]:
            log(f"  conv M={shape[0]:8d} N={shape[1]:4d} K={shape[2]:5d} k{shape[3]} {('f32in ', 'planes', 'strip ', 'tokmlp')[shape[4]]} x{n:4d}: "
                f"{ms:7.1f} ms {1e3 * ms / n:7.1f} us {f / ms / 1e9:6.1f} TFLOP/s {nb / ms / 1e9:5.2f} TB/s (algorithmic bytes)")
        lp = ops.pad_reflect(lrs[0], (h + 15) // 16 * 16, (w + 15) // 16 * 16)
        for name, fn in (("drct", eng.drct), ("grl", eng.grl), ("nafnet", eng.nafnet), ("mamba", eng.mamba)):
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            fn(lp)
            torch.cuda.synchronize(device)
            log(f"  {name:7s} {1e3 * (time.perf_counter() - t2):8.1f} ms")
        log(f"  conv/GEMM kernel: {len(prof)} launches, {flops / 1e12:.2f} TFLOP, {conv_s * 1e3:.1f} ms -> {achieved:.1f} TFLOP/s; "
            f"whole step {step_s * 1e3:.1f} ms")
    # ---- the dominant kernel FAMILY: conv / GEMM launches are grouped by the kernel that served them
    # (prof entry: shape[4] = 1 for the pre-split-input planes GEMM, 2 for its 3x3 tap-strip variant, 3 for the fused token chain, 0 for the fp32-input kernel)
    fams = {}
    for (e0, e1, f, shape, nbytes), ms in zip(prof, dur_ms):
        t = fams.setdefault(int(shape[4]), [0, 0.0, 0.0, 0.0])
        t[0] += 1; t[1] += ms; t[2] += f; t[3] += nbytes
    if ops.GEMM_MODE == "bf16x3":
        names = {1: ("conv_gemm_planes_kernel", "conv_gemm_planes_kernel (implicit GEMM, operands pre-split into bf16 hi/lo planes, "
                     "LDS-DMA staging; products as 3-term split-bf16 MFMA, fp32 accumulate)"),
                 2: ("conv3_strip_planes_kernel", "conv3_strip_planes_kernel (3x3 implicit GEMM on pre-split planes; the three "
                     "horizontal taps share one LDS-DMA-staged strip of A rows; 3-term split-bf16 MFMA, fp32 accumulate)"),
                 0: ("conv_gemm_bf16x3_v3_kernel", "conv_gemm_bf16x3_v3_kernel (implicit GEMM; fp32 operands split on the fly, "
                     "3-term split-bf16 MFMA, fp32 accumulate)"),
                 3: ("tok_chain_kernel", "tok_chain_kernel (token-stationary fused LayerNorm + fc1 + GELU + fc2 + residual: rows "
                     "in registers, hidden layer never leaves them, weights streamed fragment-major through LDS; 3-term "
                     "split-bf16 MFMA 16x16x32, fp32 accumulate)")}
        # 3 bf16 MFMAs per algorithmic fp32 multiply-add (1 in the plain-bf16 option)
        mfma_peak = MFMA_BF16_PEAK_TFLOPS / (1.0 if ops.gemm_mode_name() == "bf16" else 3.0)
    else:
        names = {0: ("conv_gemm_kernel", "conv_gemm_kernel (f32-input MFMA implicit GEMM)")}
        mfma_peak = MFMA_F32_PEAK_TFLOPS
    dom = max(fams, key=lambda k: fams[k][1])
    n_l, dom_ms, dom_flops, dom_bytes = fams[dom]
    ksym, kname = names[dom]
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if (h, w) == (H_LR, W_LR) and args.batch == 1 and os.path.exists(tfile):
        # PMC counters cannot be read from inside this process: the per-launch HBM bytes come from the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/pmc_traffic.py; corrected as the
        # guide prescribes: FETCH_SIZE x2 on gfx950)
        ent = json.load(open(tfile)).get(ksym)
        if ent:
            traffic, traffic_src = ent["hbm_bytes_per_launch"], "profiles/r03_pmc_traffic.json"
    mean_s = dom_ms / 1e3 / n_l
    bytes_l, flops_l = dom_bytes / n_l, dom_flops / n_l
    gbps = bytes_l / mean_s / 1e9
    dom_tflops = dom_flops / (dom_ms / 1e3) / 1e12
    # the binding roofline of the average launch = the larger of its two lower bounds
    t_mfma, t_hbm = flops_l / (mfma_peak * 1e12), bytes_l / (HBM_PEAK_GBPS * 1e9)
    if t_hbm >= t_mfma:
        roofline = {"kernel": kname, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBPS}
    else:
        roofline = {"kernel": kname, "bound": "mfma", "achieved": dom_tflops, "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": dom_tflops / mfma_peak}
    roofline.update({"traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_l,
                     "algorithmic_flops_per_launch": flops_l, "launches_per_step": n_l, "mean_launch_us": mean_s * 1e6,
                     "mfma_view": {"achieved_tflops": dom_tflops, "peak_tflops": mfma_peak, "frac": dom_tflops / mfma_peak},
                     "hbm_view": {"achieved_gbps": gbps, "peak_gbps": HBM_PEAK_GBPS, "frac": gbps / HBM_PEAK_GBPS},
                     "kernel_share_of_step": dom_ms / 1e3 / step_s,
                     "all_conv_gemm_launches": {"launches": len(prof), "tflop": flops / 1e12, "ms": conv_s * 1e3,
                                                "tflops": achieved, "share_of_step": conv_s / step_s,
                                                "by_kernel": {names[k][0]: {"launches": v[0], "ms": v[1], "tflops": v[2] / v[1] / 1e9}
                                                              for k, v in fams.items()}}})

    if rank == 0:
        line = {"metric": f"SR output megapixels/s (x4, {w}x{h} LR -> {w * SCALE}x{h * SCALE}, full 4-expert + 7-phase fusion)",
                "value": value, "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": ("f32" if ops.GEMM_MODE == "f32" else
                          "bf16 operands, f32 accumulate and storage (precision option, not the default)" if ops.gemm_mode_name() == "bf16" else
                          "f32 (GEMM/conv products as 3-term split-bf16 MFMA, fp32 accumulate)"), "data": "synthetic (seeded LR images; random-init weights of the exact architecture)",
                "config": {"workload": f"CompleteEnhancedFusionSR hot path: {w}x{h} LR image per step per GPU "
                                       f"(pad16 -> DRCT-L + GRL-B + NAFNet-w64 + MambaIR -> fusion), batch {args.batch}",
                           "lr_hw": [h, w], "images_per_step_per_gpu": args.batch, "parallelism": f"image-parallel x{world}",
                           "hip_graph": bool(args.graph),
                           "small_experts": bool(args.small)},
                "gemm_mode": ops.gemm_mode_name(), "roofline": roofline}
        default_geometry = (h, w) == (H_LR, W_LR) and args.batch == 1 and not args.small and not args.graph
        if world == 1 and default_geometry and not args.no_extras:
            # ---- north_star's second geometry and BASELINE config 5, measured in the same process after the headline
            # (value / metric / config above are unaffected)
            log("extras: 64x64 LR tile (batch 1, HIP-graph replay = the engine's default for small tiles) ...")
            t64 = E.nchw_to_map(synth_lr(4321, 64, 64), device)
            for _ in range(3):
                eng.process(t64)
            torch.cuda.synchronize(device)
            n64 = 30
            t1 = time.perf_counter()
            for _ in range(n64):
                o64 = eng.process(t64)
            torch.cuda.synchronize(device)
            ms64 = 1e3 * (time.perf_counter() - t1) / n64
            tfl64 = 884.0e9 / 1e12                      # 215.8 MFLOP per LR pixel = 884 GFLOP per 64x64 tile
            line["tile64"] = {"ms": ms64, "mp_s": 256 * 256 / 1e6 / (ms64 / 1e3), "tflops": tfl64 / (ms64 / 1e3),
                              "frac": tfl64 / (ms64 / 1e3) / (MFMA_BF16_PEAK_TFLOPS / 3.0 if ops.GEMM_MODE != "f32" else MFMA_F32_PEAK_TFLOPS),
                              "frac_of": "split-bf16 MFMA ceiling (2500 / 3 TFLOP/s)" if ops.GEMM_MODE != "f32" else "f32 MFMA peak",
                              "workload": "one 64x64 LR tile -> 256x256, full 4-expert + fusion path, batch 1, graph replay",
                              "steps": n64}
            del eng
            torch.cuda.empty_cache()
            log("extras: BASELINE config 5, 3 training steps at B = 32 ...")
            tl = train_measure(args, steps=3, warmup=1, cpu=False, quiet=True)
            line["train_step"] = {"ms": tl["ms_per_step"], "patches_s": tl["config"]["patches_per_s"], "frac": tl["roofline"]["frac"],
                                  "frac_of": "split-bf16 MFMA ceiling, algorithmic 3 x forward FLOPs", "steps": 3, "batch": 32,
                                  "attention_dropout": tl["config"]["attention_dropout"], "final_loss": tl["final_loss"]}
        if world == 1 and not args.no_cpu_baseline:
            naf_cfg = dict(enc_blks=(1, 1, 1, 1), mid_blks=1, dec_blks=(1, 1, 1, 1)) if args.small else None
            log("cpu_baseline: oracle on 64x64 tiles ...")
            line["cpu_baseline"] = cpu_baseline(weights, naf_cfg, full=args.cpu_full)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            line["gpu_over_cpu_at_metric_size"] = value / line["cpu_baseline"]["at_metric_size"]["value"]
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
