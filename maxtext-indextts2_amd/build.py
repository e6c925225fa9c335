"""Build recipe for libixt.so (HIP, gfx950 only): `python -m ...build` or build.build()."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libixt.so")
SOURCES = ["ixt_api.hip"]
HEADERS = ["ixt_device.h", "ixt_kernels.h", "ixt_gemma_kernels.h", "ixt_moe_kernels.h", "ixt_gemm.h", "ixt_small.h", "ixt_fp8.h", "ixt_train_kernels.h", "ixt_train.hip", os.path.join("..", "..", "include", "ixt.h"),
           os.path.join("..", "..", "include", "ixt_train.h")]

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


def source_hash() -> str:
  """sha256 over the sources, headers and compiler flags the library is built from.  It is compiled into libixt.so
  (`ixt_build_id()`); `_lib.load()` recomputes it and refuses a binary built from other sources."""
  h = hashlib.sha256()
  for name in SOURCES + HEADERS:
    with open(os.path.join(CSRC, name), "rb") as f:
      h.update(name.encode() + b"\0" + f.read() + b"\0")
  h.update(" ".join(HIPCC_FLAGS).encode())
  return h.hexdigest()[:32]


def built_hash() -> str:
  """The hash recorded next to the library by the last build ("" if none)."""
  try:
    with open(LIB + ".hash", "r", encoding="utf-8") as f:
      return f.read().strip()
  except OSError:
    return ""


def needs_build() -> bool:
  return not os.path.exists(LIB) or built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
  if not force and not needs_build():
    return LIB
  h = source_hash()
  cmd = [hipcc()] + HIPCC_FLAGS + [f'-DIXT_SOURCE_HASH="{h}"'] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.check_call(cmd)
  with open(LIB + ".hash", "w", encoding="utf-8") as f:
    f.write(h + "\n")
  return LIB


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
