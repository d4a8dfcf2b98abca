"""Turn a rocprofv3 --kernel-trace --stats CSV into the summary committed under profiles/.
usage: python tools/summarize_prof.py gpurun_out/prof_x/<host>/<pid>_kernel_stats.csv profiles/rNN_name.md "<command>" """
import csv
import sys


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
        f.write(f"total GPU kernel time: {tot / 1e6:.1f} ms over all dispatches of the run\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "/")
            if len(name) > 90:
                name = name[:87] + "..."
            f.write(f"| `{name}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main()
