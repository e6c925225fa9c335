#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv (kernel names can be kilobytes long) into a short table."""
import csv
import re
import sys


def short(name: str) -> str:
  name = re.sub(r"\(.*$", "", name)
  name = name.replace("void ", "")
  return name[:70]


def main(path, steps=None):
  rows = list(csv.DictReader(open(path)))
  tot = sum(float(r["TotalDurationNs"]) for r in rows)
  print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
  for r in rows:
    t = float(r["TotalDurationNs"])
    if t / tot < 0.002:
      continue
    print(f"{short(r['Name']):70s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.2f} {t/1e6:9.3f} {100*t/tot:6.2f}")
  print(f"total kernel time {tot/1e6:.3f} ms")


if __name__ == "__main__":
  main(sys.argv[1])
