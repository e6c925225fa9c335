"""Build recipe for libixt.so (HIP, gfx950 only): `python -m ...build` or build.build()."""
from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libixt.so")
SOURCES = ["ixt_api.hip"]
HEADERS = ["ixt_device.h", "ixt_kernels.h", os.path.join("..", "..", "include", "ixt.h")]

# -ffp-contract=off: the numerics contract allows fused multiply-adds only where the source
# says fmaf; correctly rounded f32 divide/sqrt and IEEE denormals are hipcc defaults and are stated explicitly.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function",
]


def hipcc() -> str:
  for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
    if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
      return cand
  raise RuntimeError("hipcc not found")


def needs_build() -> bool:
  if not os.path.exists(LIB):
    return True
  t = os.path.getmtime(LIB)
  deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
  return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
  if not force and not needs_build():
    return LIB
  cmd = [hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.check_call(cmd)
  return LIB


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
