"""Where the start-up time of io.main goes: template init, checkpoint read, engine construction (weight packing), first image."""
import cProfile, importlib, os, pstats, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W = importlib.import_module("image-super-resolution_amd.weights")
E = importlib.import_module("image-super-resolution_amd.engine")
import bench

dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
t = time.perf_counter(); w = W.random_weights(); print(f"random_weights {time.perf_counter() - t:.2f} s", flush=True)
t = time.perf_counter(); shapes = W.random_weights(shapes_only=True); print(f"random_weights(shapes_only) {time.perf_counter() - t:.2f} s", flush=True)
with tempfile.TemporaryDirectory() as d:
    W.save_model_dir(d, w)
    t = time.perf_counter(); w2 = W.load_model_dir(d, shapes); print(f"load_model_dir {time.perf_counter() - t:.2f} s", flush=True)
pr = cProfile.Profile()
t = time.perf_counter(); pr.enable(); eng = E.Engine(w2, dev); torch.cuda.synchronize(); pr.disable()
print(f"Engine() {time.perf_counter() - t:.2f} s", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
lr = E.nchw_to_map(bench.synth_lr(1, 340, 510), dev)
for i in range(3):
    t = time.perf_counter(); eng.process(lr); torch.cuda.synchronize(); print(f"image {i}: {time.perf_counter() - t:.3f} s", flush=True)
