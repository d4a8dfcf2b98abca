"""Per-operator GPU time of one 510x340 pipeline pass, by C-ABI entry point and shape.

Every hip.call is bracketed by two events on the current stream (experts run one after the other so that nothing
overlaps); the key is (expert, entry point, the small integer arguments = dimensions / strides / flags).  Prints the
entries sorted by total time -- the table that says WHICH GEMM shapes / elementwise passes are worth fusing next.

    python tools/op_times.py [--top 60] [--h 340 --w 510] [--by-entry]
"""
import argparse
import collections
import importlib
import os
import sys

import torch

os.environ.setdefault("FFSR_CONCURRENT_EXPERTS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = lambda m: importlib.import_module("image-super-resolution_amd." + m)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--h", type=int, default=340)
    ap.add_argument("--w", type=int, default=510)
    ap.add_argument("--by-entry", action="store_true")
    ap.add_argument("--gemm", default=None)
    args = ap.parse_args()
    hip, W, E, ops = pkg("hip"), pkg("weights"), pkg("engine"), pkg("ops")
    if args.gemm:
        ops.set_gemm_mode(args.gemm)
    dev = torch.device("cuda:0")
    eng = E.Engine(W.random_weights(seed=0), dev)
    lr = torch.rand(1, args.h, args.w, 3, device=dev)
    eng.process(lr, graph=False)
    torch.cuda.synchronize()

    records, tag = [], ["fusion"]
    real_call = hip.call

    def timed(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *a)
        e1.record()
        dims = tuple(int(v) for v in a if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 22))
        records.append((tag[0], name, dims, e0, e1))

    hip.call = timed
    def wrap(fn, nm):
        def inner(*a, **k):
            prev, tag[0] = tag[0], nm
            try:
                return fn(*a, **k)
            finally:
                tag[0] = prev
        return inner
    for name in ("drct", "grl", "nafnet", "mamba"):             # __call__ is looked up on the type
        cls = type(getattr(eng, name))
        cls.__call__ = wrap(cls.__call__, name)
    eng.process(lr, graph=False)
    torch.cuda.synchronize()
    hip.call = real_call

    agg = collections.defaultdict(lambda: [0, 0.0])
    per_expert = collections.defaultdict(float)
    for t, name, dims, e0, e1 in records:
        ms = e0.elapsed_time(e1)
        key = (t, name) if args.by_entry else (t, name, dims)
        agg[key][0] += 1
        agg[key][1] += ms
        per_expert[t] += ms
    total = sum(v[1] for v in agg.values())
    print(f"{len(records)} calls, {total:.1f} ms between events (includes launch gaps of tiny kernels)")
    print("  ".join(f"{k} {v:.1f}" for k, v in sorted(per_expert.items(), key=lambda kv: -kv[1])))
    for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{ms:8.2f} ms  {n:5d} x {1e3 * ms / n:8.1f} us  {key[0]:7s} {key[1]:34s} {key[2] if len(key) > 2 else ''}")


if __name__ == "__main__":
    main()
