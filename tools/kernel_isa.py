#!/usr/bin/env python3
"""Instruction mix of the kernels in libatx.so whose demangled name matches a pattern, read from the embedded code objects
(no GPU):   python tools/kernel_isa.py 'combine_kernel<double.*, 1>' [--lib PATH] [--dump]
Counts are static (whole kernel body, every branch): a guide to what a wave can issue at most, next to tools/kernel_resources.py.
A float64 streaming kernel on MI355X has room for ~100 VALU instructions per element before arithmetic takes as long as HBM
(16 lanes per SIMD and cycle: one wave-wide instruction every 4 cycles; DESIGN.md §3 "Where the transcendental operators spend their time")."""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)")


def code_objects(lib: str, tmp: str) -> list[str]:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    objs = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, obj = os.path.join(tmp, f"bundle{n}.bin"), os.path.join(tmp, f"device{n}.co")
        open(piece, "wb").write(blob[a:b])
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={piece}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={obj}"], check=True, capture_output=True)
        objs.append(obj)
    return objs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern")
    ap.add_argument("--lib", default=os.path.join(ROOT, "anemoi-transform_amd", "lib", "libatx.so"))
    ap.add_argument("--dump", action="store_true", help="print the disassembly of the matching kernels")
    args = ap.parse_args()
    pat = re.compile(args.pattern)
    with tempfile.TemporaryDirectory() as tmp:
        for obj in code_objects(args.lib, tmp):
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", "--no-show-raw-insn", obj], check=True, capture_output=True,
                                  text=True).stdout
            for block in re.split(r"\n(?=[0-9a-f]{16} <)", text):
                head = re.match(r"[0-9a-f]{16} <(.*)>:", block)
                if not head or not pat.search(head.group(1)) or head.group(1).endswith(".kd"):
                    continue
                ops = [ln.split()[0] for ln in block.splitlines()[1:] if ln.strip() and not ln.lstrip().startswith(("<", "/"))]
                c = Counter(ops)
                valu = sum(n for k, n in c.items() if k.startswith("v_"))
                print(f"{head.group(1)[:200]}\n   instructions {len(ops)}: VALU {valu} (f64 {sum(n for k, n in c.items() if 'f64' in k)}, "
                      f"v_mov {sum(n for k, n in c.items() if k.startswith('v_mov'))}, cndmask {sum(n for k, n in c.items() if k.startswith('v_cndmask'))}, "
                      f"quarter-rate {sum(n for k, n in c.items() if TRANS.match(k))}), SALU {sum(n for k, n in c.items() if k.startswith('s_'))}, "
                      f"global {sum(n for k, n in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))}, LDS {sum(n for k, n in c.items() if k.startswith('ds_'))}")
                if args.dump:
                    print(block)


if __name__ == "__main__":
    sys.exit(main())
