#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a short table.
usage: summarize_rocprof.py <kernel_stats.csv> [top_n]"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    base = m.group(1) if m else name
    return base[:90]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"# total device time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows[:top]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main()
