"""Experiment: capture Engine.process into a HIP graph (torch.cuda.CUDAGraph) and replay it. usage: graph_try.py H W [steps]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W_ = importlib.import_module("image-super-resolution_amd.weights")
E = importlib.import_module("image-super-resolution_amd.engine")
sys.path.insert(0, ROOT)
import bench
h, w = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
eng = E.Engine(W_.random_weights(seed=0), dev)
lr = E.nchw_to_map(bench.synth_lr(1234, h, w, 1), dev)
for _ in range(2):
    ref = eng.process(lr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = eng.process(lr)
torch.cuda.synchronize()
print(f"eager : {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.process(lr)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    gout = eng.process(lr)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph vs eager max diff:", (gout - ref).abs().max().item())
t0 = time.perf_counter()
for _ in range(steps):
    g.replay()
torch.cuda.synchronize()
print(f"graph : {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
