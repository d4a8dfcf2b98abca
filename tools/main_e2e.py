"""End-to-end throughput of the drop-in entry point models/team29_FreqFusionSR/io.py::main (PNG decode -> H2D -> engine
-> D2H -> PNG encode) on synthetic 510x340 images.  usage: python tools/main_e2e.py [n_images]
main() itself prints the wall time and rate of its image loop (decode / encode overlapped on worker threads)."""
import importlib, os, sys, tempfile, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image
W = importlib.import_module("image-super-resolution_amd.weights")
from models.team29_FreqFusionSR import main
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with tempfile.TemporaryDirectory() as tmp:
    model_dir = os.path.join(tmp, "model_zoo")
    t0 = time.perf_counter()
    W.save_model_dir(model_dir, W.random_weights(seed=0))
    print(f"wrote full-size checkpoints in {time.perf_counter() - t0:.1f} s", flush=True)
    times = {}
    for count in (n,):
        inp, out = os.path.join(tmp, f"in{count}"), os.path.join(tmp, f"out{count}")
        os.makedirs(inp)
        for i in range(count):
            lr = bench.synth_lr(100 + i, 340, 510)[0].permute(1, 2, 0).mul(255).round().byte().numpy()
            Image.fromarray(lr).save(os.path.join(inp, f"{i:04d}.png"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        main(model_dir=model_dir, input_path=inp, output_path=out, device=torch.device("cuda"))
        torch.cuda.synchronize()
        times[count] = time.perf_counter() - t0
        assert len(os.listdir(out)) == count
    print(f"main(): {times[n]:.2f} s for {n} images including the checkpoint load; the image-loop rate is printed by main() above")
