"""Turn the raw dump of tools/mfma_dump.hip (run on an MI355X: `/tmp/mfma_dump gpurun_out/mfma_hw.bin`) into the committed
fixture tests/golden/mfma_step_hw.npz: inputs and the HARDWARE's outputs of v_mfma_f32_16x16x32_bf16, every third case.

  python tests/golden/make_mfma_golden.py gpurun_out/mfma_hw.bin
"""
import sys
import numpy as np


def main(path):
  raw = open(path, "rb").read()
  n = int(np.frombuffer(raw, np.int32, 1)[0])
  off = 4
  a = np.frombuffer(raw, np.uint16, n * 32, off).reshape(n, 32); off += n * 64
  b = np.frombuffer(raw, np.uint16, n * 32, off).reshape(n, 32); off += n * 64
  c = np.frombuffer(raw, np.float32, n, off); off += n * 4
  d = np.frombuffer(raw, np.float32, n, off)
  # keep every class (the dump cycles through 8 generators); 3 and 8 are coprime, so a stride of 3 keeps all of them
  sel = np.arange(0, n, 3)
  np.savez_compressed("tests/golden/mfma_step_hw.npz", a=a[sel], b=b[sel], c=c[sel].view(np.uint32), d=d[sel].view(np.uint32),
                      source=np.array("v_mfma_f32_16x16x32_bf16 on MI355X (gfx950), tools/mfma_dump.hip seed 2026"))
  print("cases", sel.size)


if __name__ == "__main__":
  main(sys.argv[1])
