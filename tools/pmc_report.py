"""Derived MFMA / LDS metrics from tools/gemm_pmc.sh's summary + a rocprofv3 --stats CSV of the same command.
usage: python tools/pmc_report.py gpurun_out/gpmc_<tag>_summary.txt gpurun_out/prof_<tag>_seq_kernel_stats.csv profiles/<out>.md"""
import csv
import re
import sys

SIMDS, SE = 1024, 32          # MI355X: 256 CUs x 4 SIMDs; SQ_BUSY_CYCLES is summed over the 32 shader engines


def main():
    summ, stats, out = sys.argv[1:4]
    dur = {}
    for r in csv.DictReader(open(stats)):
        dur[r["Name"].replace("(anonymous namespace)::", "")[:70]] = float(r["AverageNs"])
    blocks, cur = {}, None
    for ln in open(summ):
        m = re.match(r"^void (.*?) dispatches (\d+)", ln.replace("(anonymous namespace)::", ""))
        if m:
            cur = blocks.setdefault(m.group(1)[:70], {"dispatches": int(m.group(2))})
        elif cur is not None and ln.strip():
            k, v = ln.split()[:2]
            cur[k] = float(v)
    with open(out, "w") as f:
        f.write("# MFMA-busy and LDS counters of the conv / GEMM kernels (rocprofv3 --pmc, one 340x510 image)\n\n"
                "command: `bash tools/gemm_pmc.sh` = two passes of `rocprofv3 --pmc <8 SQ counters> -- python3 bench.py --steps 1 --warmup 0 "
                "--no-cpu-baseline` (no other trace domains).  Values are per dispatch.  `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs "
                "/ (SQ_BUSY_CYCLES / 32 shader engines) = fraction of the kernel's cycles in which a SIMD's matrix pipe is executing; "
                "`issue stall` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, `parked` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier), "
                "`LDS conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n"
                "| kernel | dispatches | MFMA busy | issue stall | parked | active | LDS conflict | VMEM rd / wr instr |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for name, c in blocks.items():
            if "SQ_BUSY_CYCLES" not in c:
                continue
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / (c["SQ_BUSY_CYCLES"] / SE)
            wc = c["SQ_WAVE_CYCLES"]
            conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
            f.write(f"| `{name}` | {c['dispatches']} | {busy:.2f} | {c['SQ_WAIT_INST_ANY'] / wc:.2f} | {c['SQ_WAIT_ANY'] / wc:.2f} | "
                    f"{c['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {conf:.3f} | {c.get('SQ_INSTS_VMEM_RD', 0):.0f} / {c.get('SQ_INSTS_VMEM_WR', 0):.0f} |\n")


if __name__ == "__main__":
    main()
