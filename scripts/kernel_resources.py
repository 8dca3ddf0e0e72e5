"""Registers, spills, scratch and static LDS of every kernel in a built liblbdrn_hip*.so, read from the code object's
metadata note (no GPU needed).

    python scripts/kernel_resources.py [path/to/lib.so] [name-substring ...]

Also imported by tests/test_kernel_disassembly.py (the shipped library must hold no kernel with scratch)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def kernel_resources(so):
    """[{name, demangled, vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, scratch, lds}] for the gfx950 code objects in `so`."""
    work = tempfile.mkdtemp(prefix="lbdrn_co_")
    try:
        shutil.copy(so, os.path.join(work, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
        res = []
        for co in sorted(f for f in os.listdir(work) if f.startswith("lib.so.") and f.endswith("gfx950")):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(work, co)],
                                   capture_output=True, text=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
                blk = "    .agpr_count:" + blk

                def field(key, default=0):
                    m = re.search(r"^    \." + key + r":\s*(\S+)", blk, re.M)   # (kernel level: four spaces; arguments sit deeper)
                    return m.group(1) if m else default
                name = field("name", "")
                if not name or name == "0":
                    continue
                res.append(dict(name=name, vgpr=int(field("vgpr_count")), agpr=int(field("agpr_count")),
                                sgpr=int(field("sgpr_count")), spill_vgpr=int(field("vgpr_spill_count")),
                                spill_sgpr=int(field("sgpr_spill_count")), scratch=int(field("private_segment_fixed_size")),
                                lds=int(field("group_segment_fixed_size"))))
        for r, d in zip(res, demangle([r["name"] for r in res])):
            r["demangled"] = re.sub(r"^void ", "", d)
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    so = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "lbdrn-msic_amd", "liblbdrn_hip.so")
    rows = kernel_resources(so)
    if args:
        rows = [r for r in rows if any(a in r["demangled"] for a in args)]
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spillV':>6} {'scratch':>7} {'lds':>7}  kernel")
    for r in sorted(rows, key=lambda r: r["demangled"]):
        print(f"{r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['spill_vgpr']:6d} {r['scratch']:7d} {r['lds']:7d}  {r['demangled'][:150]}")
