"""Where OpenJPEG spends the encode of a tile's MSB planes (the reference's MSB payload: gdal_translate -of JP2OpenJPEG,
ref encode.py:137) -- the figure a GPU JPEG 2000 coder would have to beat.  Host only, no GPU:
scripts/jp2_profile.c samples the program counter at 1 kHz around ONE single-threaded lbdrn_jp2_encode of the K = 5 planes
of the synthetic tile; this script resolves the samples against libopenjp2's own symbol table and sums them by stage.
    python scripts/jp2_profile.py [bands=8] [side=2048]"""
import bisect
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lbdrn-msic_amd"))
from lbdrn_hip.synth import synthetic_tile  # noqa: E402

bands = int(sys.argv[1]) if len(sys.argv) > 1 else 8
side = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
lib = os.path.join(ROOT, "lbdrn-msic_amd", "liblbdrn_jp2.so")
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, "jp2_profile")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "scripts", "jp2_profile.c"), "-ldl"])
    msb = (synthetic_tile(0, bands, side, side) >> 5).astype(np.uint16)
    bits = 8 if int(msb.max()) <= 255 else 16
    raw = os.path.join(d, "planes.raw")
    msb.tofile(raw)
    out = os.path.join(d, "out.txt")
    subprocess.check_call([exe, lib, raw, str(bands), str(side), str(side), str(bits), out])
    text = open(out).read().splitlines()
head = text[0]
maps = []     # (lo, hi, file offset, path)
for ln in text:
    if ln.startswith("map "):
        m = re.match(r"map ([0-9a-f]+)-([0-9a-f]+) \S+ ([0-9a-f]+) \S+ \S+\s+(\S+)", ln)
        if m:
            maps.append((int(m.group(1), 16), int(m.group(2), 16), int(m.group(3), 16), m.group(4)))
pcs = [int(ln[3:], 16) for ln in text if ln.startswith("pc ")]
symtab = {}


def symbols(path):
    if path not in symtab:
        syms = []
        for ln in subprocess.run(["nm", "-n", "--defined-only", path], capture_output=True, text=True).stdout.splitlines():
            p = ln.split()
            if len(p) == 3 and p[1] in "tTwW":
                syms.append((int(p[0], 16), p[2]))
        if not syms:
            for ln in subprocess.run(["nm", "-n", "-D", "--defined-only", path], capture_output=True, text=True).stdout.splitlines():
                p = ln.split()
                if len(p) == 3 and p[1] in "tTwWi":
                    syms.append((int(p[0], 16), p[2]))
        symtab[path] = (sorted(syms), [a for a, _ in sorted(syms)])
    return symtab[path]


def first_load_base(path):
    return min(lo - off for lo, hi, off, p in maps if p == path)


by_sym = {}
for pc in pcs:
    name = "?"
    for lo, hi, off, path in maps:
        if lo <= pc < hi and path.startswith("/"):
            syms, addrs = symbols(path)
            rel = pc - first_load_base(path)
            k = bisect.bisect_right(addrs, rel) - 1
            name = (syms[k][1] if k >= 0 else "?") + "@" + os.path.basename(path)
            break
    by_sym[name] = by_sym.get(name, 0) + 1


def stage(name):
    n = name.split("@")[0]
    if n.startswith("opj_dwt"):
        return "DWT (5/3 lifting, deinterleave)"
    if n.startswith("opj_t1") or n.startswith("opj_mqc"):
        return "tier-1 (bit-plane context modelling + MQ arithmetic coder)"
    if n.startswith("opj_t2") or n.startswith("opj_bio") or n.startswith("opj_tgt") or n.startswith("opj_pi"):
        return "tier-2 (packet headers, tag trees)"
    if n.startswith("opj_tcd") or n.startswith("opj_mct") or n.startswith("opj_j2k") or n.startswith("opj_jp2") or n.startswith("opj_"):
        return "tile set-up, DC shift, rate allocation, stream"
    return "other (memcpy / memset, the shim, libc)"


total = len(pcs)
by_stage = {}
for name, c in by_sym.items():
    by_stage[stage(name)] = by_stage.get(stage(name), 0) + c
print(f"{bands} x {side} x {side} planes of the synthetic tile at K = 5 ({bits}-bit), one thread: {head}")
for s, c in sorted(by_stage.items(), key=lambda kv: -kv[1]):
    print(f"  {100.0 * c / total:5.1f} %  {s}")
print("  top symbols:")
for name, c in sorted(by_sym.items(), key=lambda kv: -kv[1])[:12]:
    print(f"    {100.0 * c / total:5.1f} %  {name}")
