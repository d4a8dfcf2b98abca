"""Timeline of ONE graph replay from a rocprofv3 --kernel-trace CSV: per hardware queue the busy time and the idle gaps, and the
kernels that cover the replay's span.   python tools/graph_timeline.py <dir with *_kernel_trace.csv> [replay index from the end]"""
import collections
import csv
import glob
import sys


def main():
    path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    # replays are separated by long idle gaps (host sync between steps): split where the gap to the previous END exceeds 200 us
    groups, cur, last_end = [], [], None
    for s, e, q, n in rows:
        if last_end is not None and s - last_end > 200_000 and cur:
            groups.append(cur)
            cur = []
        cur.append((s, e, q, n))
        last_end = e if last_end is None else max(last_end, e)
    groups.append(cur)
    g = groups[-back]
    t0, t1 = min(r[0] for r in g), max(r[1] for r in g)
    print(f"{len(groups)} groups; analysing group of {len(g)} kernels, span {(t1 - t0) / 1e6:.3f} ms")
    byq = collections.defaultdict(list)
    for s, e, q, n in g:
        byq[q].append((s, e, n))
    for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        print(f"queue {q}: {len(ks)} kernels, busy {busy / 1e6:.3f} ms, first {(ks[0][0] - t0) / 1e6:.3f} last end {(max(e for _, e, _ in ks) - t0) / 1e6:.3f} ms")
    # union coverage: time with >= 1 kernel running, and the concurrency histogram
    ev = []
    for s, e, q, n in g:
        ev += [(s, 1), (e, -1)]
    ev.sort()
    depth, prev, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[depth] += t - prev
        prev, depth = t, depth + d
    print("concurrency histogram (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
    agg = collections.Counter()
    for s, e, q, n in g:
        agg[n.split("(")[0][:80]] += e - s
    for n, t in agg.most_common(25):
        print(f"  {t / 1e6:7.3f} ms  {n}")


if __name__ == "__main__":
    main()
