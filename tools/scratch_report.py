#!/usr/bin/env python3
"""Which kernels of the built libsymode_hip.so touch scratch memory?

Pulls every gfx950 code object out of the library's clang offload bundles (section .hip_fatbin), disassembles it with
llvm-objdump and counts scratch_* instructions per kernel.  `--fail-on PATTERN` exits 1 if a kernel whose demangled name
matches the regular expression has any (tests/test_abi.py uses it for the D <= 3, order <= 3 libraries).

    python tools/scratch_report.py [path/to/libsymode_hip.so] [--fail-on 'Library<[123], [123], ']
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(blob):
    """yield (triple, bytes) of every device code object of every bundle in the file"""
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "amdgcn" in triple and size > 0:
                yield triple, blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def scratch_by_kernel(lib):
    blob = open(lib, "rb").read()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, (_, co) in enumerate(code_objects(blob)):
            path = os.path.join(tmp, f"{k}.co")
            with open(path, "wb") as f:
                f.write(co)
            asm = subprocess.run([OBJDUMP, "-d", "--demangle", path], capture_output=True, text=True, check=True).stdout
            name = None
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    name = m.group(1)
                    out.setdefault(name, 0)
                elif name is not None and re.search(r"\bscratch_(load|store)", line):
                    out[name] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(ROOT, "symmetry-ode-discovery_amd", "libsymode_hip.so"))
    ap.add_argument("--fail-on", default=None, help="regular expression on the demangled kernel name")
    a = ap.parse_args()
    res = scratch_by_kernel(a.lib)
    bad = {k: v for k, v in res.items() if v > 0}
    print(f"{len(res)} kernels, {len(bad)} with scratch instructions")
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1]):
        print(f"{v:6d}  {k[:150]}")
    if a.fail_on:
        hit = [k for k in bad if re.search(a.fail_on, k)]
        if hit:
            print(f"FAIL: {len(hit)} kernels matching {a.fail_on!r} use scratch")
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
