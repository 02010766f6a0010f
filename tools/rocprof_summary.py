"""Turn a rocprofv3 --kernel-trace CSV into the per-kernel summary committed under profiles/.

    python tools/rocprof_summary.py <dir with *_kernel_trace.csv> [--top 40] > profiles/<name>.md
"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import pretty  # noqa: E402


def short(name):
    return pretty(name)[:90]


def main():
    d = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    total = 0.0
    for f in files:
        for r in csv.DictReader(open(f)):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
            total += dur
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| `{k}` | {n} | {t / 1e3:.3f} | {t / n:.1f} | {mn:.1f} | {mx:.1f} | {100 * t / total:.2f} |")
    print(f"\ntotal kernel time: {total / 1e3:.3f} ms over {sum(a[0] for a in agg.values())} launches")


if __name__ == "__main__":
    main()
