#!/usr/bin/env python3
"""Micro-benchmark of the canonical DOT kernel on the decode/prefill shapes of C2 (run on the GPU box)."""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H

SHAPES = [("qkv", 2048, 6144), ("o", 2048, 2048), ("wi", 2048, 14336), ("wo", 7168, 2048), ("unembed", 2048, 264192)]


def main():
  rows = [int(a) for a in sys.argv[1:]] or [8, 1]
  cfg = H.make_config(per_device_batch_size=float(max(rows + [8])), max_prefill_predict_length=256, max_target_length=512, pagedattn_num_pages=64,
                      model_name="default", base_emb_dim=2048, base_mlp_dim=7168, base_num_decoder_layers=1, vocab_size=264192,
                      base_num_query_heads=16, base_num_kv_heads=16, head_dim=128, pagedattn_tokens_per_page=32)
  maxengine = importlib.import_module("maxtext-indextts2_amd.maxengine")
  eng = maxengine.MaxEngine(cfg)
  lib = eng.lib
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  # kernel variants: (label, {option: value}); the ring GEMM (csrc/ixt_gemm.h) is the default from 128 rows
  variants = [("default", {})]
  if max(rows) >= 128:
    variants += [("ring 128x128", {"ring_cfg": 1}), ("ring 128x64", {"ring_cfg": 2}), ("ring 64x32", {"ring_cfg": 3}), ("ring split", {"ring_split": 1}), ("ring 128x128 ring-layout W", {"ring_cfg": 1, "ring_wlayout": 1}),
                 ("ring 128x64 ring-layout W", {"ring_cfg": 2, "ring_wlayout": 1}), ("ring 64x32 ring-layout W", {"ring_cfg": 3, "ring_wlayout": 1}),
                 ("k_gemm / k_dotb (round 1)", {"ring_min_rows": 1 << 30})]
  for R in rows:
   for label, opts in (variants if R >= 128 else variants[:1]):
    for k, v in opts.items():
      H.lib.check(lib.ixt_set_option(eng._ctx, k.encode(), v))
    print(f"--- R={R} {label}", flush=True)
    for name, K, N in SHAPES:
      if R > 8 and name == "unembed":
        continue
      x = torch.randn((R, K), device="cuda").to(torch.bfloat16)
      # several weight copies so that repeated launches do not hit the 256 MiB Infinity Cache
      ncopy = max(2, int(600e6 // (K * N * 2)))
      ws = [(torch.randn((K, N), device="cuda") * 0.02).to(torch.bfloat16) for _ in range(min(ncopy, 24))]
      out = torch.empty((R, N), dtype=torch.float32, device="cuda")
      for w in ws[:2]:
        H.lib.check(lib.ixt_op_dot(eng._ctx, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()), R, K, N, st))
      torch.cuda.synchronize()
      lib.ixt_timing_enable(eng._ctx, 1)
      lib.ixt_timing_reset(eng._ctx)
      for it in range(3):
        for w in ws:
          H.lib.check(lib.ixt_op_dot(eng._ctx, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()), R, K, N, st))
      avg, n, bpl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
      lib.ixt_timing_get(eng._ctx, b"dot", ctypes.byref(avg), ctypes.byref(n), ctypes.byref(bpl))
      lib.ixt_timing_enable(eng._ctx, 0)
      wbytes = K * N * 2
      print(f"R={R:3d} {name:8s} K={K:5d} N={N:6d}  {avg.value:8.2f} us  weights {wbytes/avg.value/1e6:7.2f} TB/s  (launches {n.value})", flush=True)
      del ws
    for k in opts:
      H.lib.check(lib.ixt_set_option(eng._ctx, k.encode(), {"ring_cfg": 0, "ring_split": 0, "ring_min_rows": 128, "ring_wlayout": 0}[k]))


if __name__ == "__main__":
  main()
