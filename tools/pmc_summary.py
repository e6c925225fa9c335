#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/pmc_collect.sh into one JSON (profiles/r03_pmc_b256.json): per (kernel, grid) the mean
counter values per launch, the HBM bytes of the decode attention (FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts the
128-byte requests of 16 B/lane streaming reads as 64 bytes and must be doubled: MI355X_MICROARCH.md, "HBM") and the matrix-core
utilisation of the GEMM kernels (SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)).
The library's build id (hash of its sources) is recorded so that bench.py quotes these numbers only for the build they were measured on."""
import csv
import glob
import importlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
  name = re.sub(r"\(.*", "", name)
  return name.replace("void ", "").strip()


def main(out_dir):
  vals = defaultdict(lambda: defaultdict(list))   # (kernel, grid) -> counter -> values
  for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
      vals[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
  kernels = []
  for (k, grid), cs in sorted(vals.items()):
    e = {"kernel": k, "grid_threads": grid, "launches": max(len(v) for v in cs.values())}
    for c, v in cs.items():
      e[c] = sum(v) / len(v)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
      e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
      e["mfma_utilisation"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    kernels.append(e)
  build = importlib.import_module("maxtext-indextts2_amd.build")
  att = [e for e in kernels if e["kernel"].startswith("k_attn_decode") and "hbm_bytes_per_launch" in e]
  att.sort(key=lambda e: -e["launches"])
  gem = [e for e in kernels if "mfma_utilisation" in e and e["kernel"].startswith("k_gemm")]
  batch = None
  for path in glob.glob(os.path.join(out_dir, "*.txt")):   # the bench's own JSON line of a counter pass names the batch it ran
    for line in open(path, errors="replace"):
      if line.startswith("{") and "utterances_per_gpu" in line:
        try:
          batch = json.loads(line)["config"]["utterances_per_gpu"]
        except (ValueError, KeyError):
          pass
  out = {"batch": batch, "model": "indextts2-1b", "build_id": build.source_hash(),
         "workload": "eager decode steps at keys 761..774 per utterance (--prefill-len 768 --decode-steps 6): the key range of the middle of a default run, NOT the timed run itself",
         "attn_hbm_bytes_per_launch": att[0]["hbm_bytes_per_launch"] if att else None,
         "mfma": {f'{e["kernel"]} grid {e["grid_threads"]}': round(e["mfma_utilisation"], 4) for e in gem},
         "source": "rocprofv3 --pmc (separate passes: FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE), tools/pmc_collect.sh: "
                   "python3 bench.py --steps 1 --warmup 0 --no-graph --sync-every 1 --prefill-len 768 --decode-steps 6 (keys 761..774 per utterance); "
                   "KiB -> bytes, FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section)",
         "kernels": kernels}
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main(sys.argv[1])
