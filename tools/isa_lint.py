#!/usr/bin/env python3
"""Scan the device code of the built library for the instruction pattern that misbehaved in round 2
(tools/experiments/README.md, "concurrency: wrong rows with two streams"): a packed fp32 instruction whose LOW result takes
the HIGH register of a source pair (``v_pk_*_f32 ... op_sel:[..1..]``).  In ``mocha_body_front`` such instructions, fed from
LDS, intermittently produced 0 when another stream's plane GEMM shared the CU.  Kernels that must not contain packed fp32
instructions carry ``MOCHA_NO_PACKED_F32`` (csrc/device_utils.h); this scan is the check that the attribute took, and that no
other kernel grew the pattern.

    python tools/isa_lint.py [path/to/libmocha_hip.so]      exit code 1 when a kernel outside the allow-list has the pattern
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels allowed to keep the pattern, with the reason
ALLOW = {
    "mocha_pose_heads": "one instruction, operands from global loads (no LDS in the kernel); data-preparation path",
}
PATTERN = re.compile(r"\bv_pk_[a-z0-9_]*_f32\b.*\bop_sel:\[[01,]*1")


def scan(lib):
    """{kernel symbol: count} over every gfx950 code object bundled in ``lib``."""
    found = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)                                   # --offloading writes the bundles next to its input
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not objs:
            raise RuntimeError(f"no device code objects found in {lib}")
        for f in objs:
            text = subprocess.run([OBJDUMP, "-d", "--demangle", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            sym = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    sym = m.group(1)
                elif sym and PATTERN.search(line):
                    found[sym] = found.get(sym, 0) + 1
    return found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "mocha_sigasia2023_amd", "libmocha_hip.so")
    found = scan(lib)
    bad = {k: v for k, v in found.items() if not any(a in k for a in ALLOW)}
    for k, v in sorted(found.items()):
        print(f"{'ALLOWED' if k not in bad else 'FOUND  '} {v:4d}  {k}")
    if not found:
        print("no packed fp32 instruction with op_sel on a high register in any kernel")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
