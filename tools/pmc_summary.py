"""Per-kernel summary of the rocprofv3 --pmc passes tools/profile_bench.sh takes (FETCH_SIZE, WRITE_SIZE and the SQ_*
counters, each in its own run with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes).

    python tools/pmc_summary.py <prof dir with pmc_fetch/ pmc_write/ pmc_sq/> "<bench args>"

Prints the markdown table and writes <prof dir>/pmc_traffic.json, which tools/copy_profiles.sh installs as
profiles/pmc_traffic.json: bench.py reads `roofline.traffic` from it (keyed by vidil_gemm_kernel_name's spelling).
FETCH_SIZE is in KB and undercounts by 2x on gfx950 (guide), hence hbm bytes = (2*FETCH + WRITE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import pretty  # noqa: E402


def main():
    out, args = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                rows[pretty(r["Kernel_Name"])[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic = {}
    print("| kernel | launches | FETCH_SIZE KB/launch (x2 = bytes read, gfx950 correction) | WRITE_SIZE KB/launch | "
          "MFMA busy / (SQ_BUSY*32 SIMD/SE) | LDS conflict / active |")
    print("|---|---:|---:|---:|---:|---:|")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
        n = len(v.get("FETCH_SIZE", [])) or 1
        f = sum(v.get("FETCH_SIZE", [0])) / n
        w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [])))
        mf, sb = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), sum(v.get("SQ_BUSY_CYCLES", [0]))
        lc, la = sum(v.get("SQ_LDS_BANK_CONFLICT", [0])), sum(v.get("SQ_LDS_IDX_ACTIVE", [0]))
        util = mf / (sb * 32) if sb else 0
        print(f"| `{k}` | {n} | {f:.0f} | {w:.0f} | {util:.3f} | {lc / la if la else 0:.3f} |")
        traffic[k] = {"launches": n, "fetch_kb_raw": round(f), "write_kb": round(w),
                      "hbm_bytes_per_launch": int((2 * f + w) * 1024), "mfma_util": round(util, 3)}
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of 'bench.py {args}'; FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (HBM section); written by tools/pmc_summary.py", "kernels": traffic},
              open(out + "/pmc_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main()
