"""Build-time check on the shipped machine code: the loader wave of k_train_stream loads its pixel
indices with an inline-asm `global_load_dwordx2` and rides the destination through an inline-asm counted wait; between
the two statements the compiler is free, by the language, to copy or spill that register before the data has landed.
It does not -- and this test keeps it that way: in the disassembly of every k_train_stream instance, nothing between the
index load and the first vector-memory wait behind it touches the destination registers, and that wait's count is no
larger than the LDS-DMA requests issued in between (requests complete in order: the load is then done)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _regs(token):
    """'v5' -> {5}; 'v[4:7]' -> {4,5,6,7}"""
    m = re.fullmatch(r"v(\d+)", token)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def test_loader_wave_index_load_reaches_its_wait_untouched(tmp_path):
    so = os.path.join(ROOT, "lbdrn-msic_amd", "liblbdrn_hip.so")
    if not (os.path.exists(OBJDUMP) and os.path.exists(so)):
        pytest.skip("llvm-objdump or the library is missing")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(so, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)   # unbundles next to the input
    checked = 0
    for co in sorted(work.glob("lib.so.*gfx950")):
        syms = subprocess.run([OBJDUMP, "-t", str(co)], capture_output=True, text=True).stdout
        names = sorted({l.split()[-1] for l in syms.splitlines() if "k_train_stream" in l and " F " in l and l.split()[-1].startswith("_Z")})
        for name in names:
            dis = subprocess.run([OBJDUMP, "-d", f"--disassemble-symbols={name}", str(co)], capture_output=True, text=True).stdout
            lines = [l.split("//")[0].strip() for l in dis.splitlines() if l.startswith("\t")]
            loads = [k for k, l in enumerate(lines) if l.startswith("global_load_dwordx2")]
            assert loads, f"{name}: the loader's index load was not found"
            # the FIRST 8-byte load of the kernel is the inline-asm one; the loader wave's parting touch of the next
            # minibatch's indices is an ordinary load whose wait the compiler places itself (on either side of a branch)
            loads = loads[:1]
            for k in loads:
                dst = _regs(lines[k].split()[1].rstrip(","))
                assert len(dst) == 2, lines[k]
                dma = 0
                for l in lines[k + 1:]:
                    toks = set(re.findall(r"v\[\d+:\d+\]|v\d+", l))
                    if l.startswith("s_waitcnt") and "vmcnt" in l:
                        n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
                        assert n <= dma, f"{name}: wait for the index load allows {n} requests in flight, only {dma} were issued behind it"
                        break
                    assert not any(dst & _regs(t) for t in toks), f"{name}: `{l}` touches the index registers before their wait"
                    dma += l.startswith("global_load_lds_dwordx4")
                    assert not l.startswith(("s_endpgm", "s_cbranch", "s_branch")), f"{name}: control flow between the index load and its wait"
                else:
                    pytest.fail(f"{name}: no vector-memory wait behind the index load")
                checked += 1
    assert checked >= 4, "k_train_stream instances not found in the library's code objects"


def test_shipped_library_exports_the_c_abi_only_and_no_kernel_uses_scratch():
    """the library is a product -- `nm -D` shows the functions of include/lbdrn_hip.h and no C++ symbol
    (-fvisibility=hidden + csrc/exports.map), no kernel in it spills (k_apply_mfma<4, 2> did: 88 registers), and its
    sources read no environment variable (every A/B and timing-only switch is a -D of csrc/build.py --variant)."""
    import sys
    so = os.path.join(ROOT, "lbdrn-msic_amd", "liblbdrn_hip.so")
    if not (os.path.exists(OBJDUMP) and os.path.exists(so)):
        pytest.skip("llvm-objdump or the library is missing")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert names and all(n.startswith("lbdrn_") for n in names), [n for n in names if not n.startswith("lbdrn_")][:5]
    header = open(os.path.join(ROOT, "include", "lbdrn_hip.h")).read()
    declared = set(re.findall(r"\b(lbdrn_[a-z0-9_]+)\s*\(", header))
    assert set(names) == declared, (sorted(set(names) - declared), sorted(declared - set(names)))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from kernel_resources import kernel_resources
    rows = kernel_resources(so)
    assert len(rows) > 50
    bad = [(r["demangled"], r["spill_vgpr"], r["scratch"]) for r in rows if r["spill_vgpr"] or r["scratch"]]
    assert not bad, bad
    csrc = os.path.join(ROOT, "lbdrn-msic_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".inc", ".hpp", ".c")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
