"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) -> JSON.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command>"
Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE x2 on gfx950 (128-B requests are
tallied at 64 B for 16-byte-per-lane streaming reads); WRITE_SIZE as is."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [set(), 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(conv_gemm_\w+_kernel|conv_gemm_kernel|conv3_strip_planes_kernel|tok_chain_kernel)", r["Kernel_Name"])
            if not m:
                continue
            a = acc[m.group(1)]
            a[0].add(r["Dispatch_Id"])
            a[1] += float(r["Counter_Value"])
    return {k: (len(v[0]), v[1]) for k, v in acc.items()}


def main():
    fd, wd, out, cmd = sys.argv[1:5]
    fetch, write = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {"command": cmd, "corrections": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane streaming reads); "
                                          "WRITE_SIZE as is; counters in KiB"}
    for k in sorted(set(fetch) | set(write)):
        nf, kf = fetch.get(k, (0, 0.0))
        nw, kw = write.get(k, (0, 0.0))
        n = max(nf, nw, 1)
        fb, wb = 2.0 * kf * 1024 / n, kw * 1024 / n
        res[k] = {"launches": n, "fetch_kib_sum": kf, "write_kib_sum": kw, "fetch_bytes_per_launch": fb,
                  "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
