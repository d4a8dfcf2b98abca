"""NTIRE submission interface -- drop-in for the reference's models/team29_FreqFusionSR/io.py.

``main(model_dir, input_path, output_path, device=None)`` keeps the reference contract (io.py:295-347; called
with keyword arguments by test.py:67): same five checkpoint files in ``model_dir``, same glob
``*.[jpJP][pnPN]*[gG]`` in sorted order, one output per input with the identical file name, uint8 RGB pixel
contract (io.py:100-120).  Underneath, every image goes through the MI355X HIP engine
(image-super-resolution_amd/engine.py); there is no CPU path.

Multi-GPU: when launched with WORLD_SIZE > 1 (torchrun, one process per GPU) rank 0 reads the checkpoints and
broadcasts them over RCCL, every rank processes ``images[rank::world]`` and joins at a barrier before
returning, so test.py's timing around ``main`` stays valid.
"""
import glob
import importlib
import os
import sys

import numpy as np
import torch
import yaml

REPO_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)

SCALE = 4
CONFIG_PATH = os.path.join(REPO_ROOT, "configs", "train_config.yaml")

__all__ = ["main"]


def _pkg(name):
    return importlib.import_module("image-super-resolution_amd." + name)


def _imread_uint(path):
    """uint8 HxWx3 RGB (cv2.imread + BGR->RGB of the reference, via PIL; grey is expanded, alpha dropped)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8)


def _imsave(img, path):
    from PIL import Image
    kw = {"compress_level": 1} if path.lower().endswith(".png") else {}     # cv2.imwrite's default PNG level
    Image.fromarray(img).save(path, **kw)


def _load_engine(model_dir, device):
    scale = SCALE
    if os.path.exists(CONFIG_PATH):
        with open(CONFIG_PATH) as f:
            cfg = yaml.safe_load(f) or {}
        scale = cfg.get("dataset", {}).get("scale", SCALE)
        fusion_cfg = cfg.get("model", {}).get("fusion", {})
        expect = {"num_experts": 4, "fusion_dim": 128, "refine_channels": 128, "refine_depth": 6, "base_channels": 64,
                  "block_size": 8}
        for k, v in expect.items():
            if fusion_cfg.get(k, v) != v:
                raise ValueError(f"configs/train_config.yaml model.fusion.{k}={fusion_cfg[k]} is not the submitted "
                                 f"architecture ({v}) this engine implements")
    weights, shard, engine = _pkg("weights"), _pkg("shard"), _pkg("engine")
    rank, world = shard.init_process_group()
    templates = weights.random_weights(shapes_only=True)      # keys + shapes; values come from the files (rank 0)
    w = weights.load_model_dir(model_dir, templates, weights.random_weights) if rank == 0 else templates
    w = shard.broadcast_weights(w, device)
    return engine.Engine(w, device, scale), rank, world


def main(model_dir, input_path, output_path, device=None):
    engine_mod = _pkg("engine")
    if device is None:
        device = torch.device("cuda")
    device = engine_mod.require_gpu(device)
    print(f"\n{'=' * 60}\n  FreqFusionSR (MI355X HIP engine)\n{'=' * 60}")
    print(f"  Weights : {model_dir}\n  Input   : {input_path}\n  Output  : {output_path}\n  Device  : {device}\n")
    eng, rank, world = _load_engine(model_dir, device)
    input_imgs = sorted(glob.glob(os.path.join(input_path, "*.[jpJP][pnPN]*[gG]")))
    os.makedirs(output_path, exist_ok=True)
    mine = _pkg("shard").shard(input_imgs, rank, world)
    print(f"  Processing {len(mine)} of {len(input_imgs)} images on rank {rank}/{world} ...")
    # host pipeline: the next image is decoded and the previous results are encoded on worker threads while the GPU
    # works on the current one; everything is joined before returning (test.py times the whole call, test.py:63-70)
    import time
    from concurrent.futures import ThreadPoolExecutor
    t_loop, out_px = time.perf_counter(), 0
    with ThreadPoolExecutor(max_workers=int(os.environ.get("FFSR_IO_THREADS", "4"))) as pool:
        saves = []
        nxt = pool.submit(_imread_uint, mine[0]) if mine else None
        for i, img_path in enumerate(mine):
            img = nxt.result()
            nxt = pool.submit(_imread_uint, mine[i + 1]) if i + 1 < len(mine) else None
            name, ext = os.path.splitext(os.path.basename(img_path))
            sr = eng.process_u8(img)
            out_px += sr.shape[0] * sr.shape[1]
            saves.append(pool.submit(_imsave, sr, os.path.join(output_path, name + ext)))
        for f in saves:
            f.result()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    t_loop = time.perf_counter() - t_loop
    print(f"  Done -- {len(mine)} images saved to {output_path} "
          f"({t_loop:.2f} s for the image loop incl. decode / encode = {out_px / 1e6 / max(t_loop, 1e-9):.2f} output-MP/s on this rank)")
