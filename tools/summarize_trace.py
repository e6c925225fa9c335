#!/usr/bin/env python3
"""Group a rocprofv3 *_kernel_trace.csv by (kernel, grid): calls, average / min duration.  Distinguishes the launches of one
templated kernel on different matrix shapes, which *_kernel_stats.csv merges."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
  name = re.sub(r"\(.*$", "", name).replace("void ", "")
  return name[:44]


def main(path):
  g = defaultdict(list)
  for r in csv.DictReader(open(path)):
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]))
    g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
  rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
  tot = sum(sum(v) for v in g.values())
  print(f"{'kernel':44s} {'grid(x,y,z)xblock':>22s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'total_ms':>9s} {'%':>6s}")
  for (name, gx, gy, gz, wx), v in rows:
    if sum(v) / tot < 0.002:
      continue
    print(f"{name:44s} {f'({gx},{gy},{gz})x{wx}':>22s} {len(v):7d} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {sum(v)/1e6:9.3f} {100*sum(v)/tot:6.2f}")
  print(f"total kernel time {tot/1e6:.3f} ms")


if __name__ == "__main__":
  main(sys.argv[1])
