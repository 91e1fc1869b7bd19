#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in libatx.so, read from the code objects embedded in the library (no GPU, no recompilation):
   python tools/kernel_resources.py [--lib PATH] [--scratch]      (--scratch: only kernels that use private scratch memory)
A kernel that indexes a private array with a run-time subscript ends up in scratch memory and loses most of its bandwidth — round 3
found the field-major per-point kernel that way (0.41 of the HBM peak instead of 0.83); tests/test_host_api.py keeps the list empty."""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_resources(lib: str) -> list[dict]:
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            piece, obj = os.path.join(tmp, f"bundle{n}.bin"), os.path.join(tmp, f"device{n}.co")
            open(piece, "wb").write(blob[a:b])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={piece}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={obj}"], check=True, capture_output=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], check=True, capture_output=True, text=True).stdout
            for block in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", notes, re.S):
                text = block.group(0)
                get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", text).group(1))  # noqa: E731
                name = re.search(r"\.name:\s+(\S+)", text).group(1)
                out.append({"name": name, "vgpr": get("vgpr_count"), "sgpr": get("sgpr_count"), "lds": get("group_segment_fixed_size"),
                            "scratch": get("private_segment_fixed_size")})
    return out


def demangle(names: list[str]) -> list[str]:
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            run = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
        except FileNotFoundError:
            continue
        if run.returncode == 0:
            return run.stdout.splitlines()
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "anemoi-transform_amd", "lib", "libatx.so"))
    ap.add_argument("--scratch", action="store_true")
    args = ap.parse_args()
    rows = kernel_resources(args.lib)
    if args.scratch:
        rows = [r for r in rows if r["scratch"] > 0]
    for r, nice in zip(rows, demangle([r["name"] for r in rows])):
        print(f"vgpr {r['vgpr']:4d} sgpr {r['sgpr']:4d} lds {r['lds']:6d} scratch {r['scratch']:5d}  {nice[:160]}")
    print(f"{len(rows)} kernels", file=sys.stderr)


if __name__ == "__main__":
    main()
