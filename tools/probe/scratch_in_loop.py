"""Where a kernel's scratch (spill) accesses sit relative to its MFMA loop: compiles one source with -save-temps into a fresh
temporary directory and lists, per kernel whose mangled name contains the filter, the scratch_* instructions between the first and
the last v_mfma.   python tools/probe/scratch_in_loop.py gemm_nt256p.hip ELi5E [-fno-honor-nans ...]"""
import os
import re
import subprocess
import sys
import tempfile

src, filt = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
csrc = os.environ.get("CSRC") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "unmore_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps", *extra, "-c",
                    os.path.join(csrc, src), "-o", "o.o"], cwd=td, check=True, capture_output=True)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(td, asm)).read()
idx = [m.start() for m in re.finditer(r"^_Z\S*:", s, flags=re.M)] + [len(s)]
for a, b in zip(idx[:-1], idx[1:]):
    f = s[a:b]
    name = f.split(":")[0]
    if filt not in name:
        continue
    lines = f.split("\n")
    sc = [i for i, l in enumerate(lines) if "scratch_" in l]
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
    if not mf:
        continue
    inloop = [i for i in sc if mf[0] <= i <= mf[-1]]
    m = re.search(r"\.vgpr_count:\s+(\d+)", f)
    print(f"{name[:90]}: {len(lines)} lines, {len(sc)} scratch ops, {len(inloop)} between the first and last MFMA ({mf[0]}..{mf[-1]})")
    for i in inloop:
        print("    ", i, lines[i].strip())
