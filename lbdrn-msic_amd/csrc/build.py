"""Build liblbdrn_hip.so for gfx950 with hipcc (in-tree, next to the Python host code).

    python lbdrn-msic_amd/csrc/build.py [--force]

-ffp-contract=off: the canonical arithmetic (lbdrn_math.hpp) spells out every fma; the compiler
must not invent more.  -fhip-fp32-correctly-rounded-divide-sqrt: IEEE division for the
normalisation p = msb/max and the sigmoid (the reference divides in numpy / torch float32).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = ["cabi.hip", "generic.hip", "apply_mfma.hip", "train_mfma.hip", "randperm.hip", "plane_codec.hip", "weights_codec.hip"]
HDRS = sorted(f for f in os.listdir(HERE) if f.endswith((".hpp", ".inc"))) + ["exports.map", "../../include/lbdrn_hip.h"]
OUT = os.path.join(os.path.dirname(HERE), "liblbdrn_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"]   # the exports are the functions include/lbdrn_hip.h declares, nothing else


JP2_OUT = os.path.join(os.path.dirname(HERE), "liblbdrn_jp2.so")


def build_jp2(force=False):
    """liblbdrn_jp2.so (include/lbdrn_jp2.h): OpenJPEG behind a C ABI, gcc, host only.  Built where openjpeg.h and
    libopenjp2 are found (this image: /opt/conda); returns None elsewhere -- the JPEG 2000 payload is then unavailable
    and says so, the default GPU payload does not need it."""
    import glob
    src = os.path.join(HERE, "jp2_shim.c")
    hdr = os.path.join(HERE, "..", "..", "include", "lbdrn_jp2.h")
    if not force and os.path.exists(JP2_OUT) and os.path.getmtime(JP2_OUT) > max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return JP2_OUT
    for root in ("/opt/conda", "/usr", "/usr/local"):
        incs = sorted(glob.glob(os.path.join(root, "include", "openjpeg-*", "openjpeg.h")))
        libs = [d for d in (os.path.join(root, "lib"), os.path.join(root, "lib", "x86_64-linux-gnu"))
                if glob.glob(os.path.join(d, "libopenjp2.so*"))]
        if incs and libs:
            lib = sorted(glob.glob(os.path.join(libs[0], "libopenjp2.so*")))[0]
            cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-Wall", "-fvisibility=hidden", "-I" + os.path.dirname(incs[-1]), "-o", JP2_OUT, src,
                   lib, "-Wl,-rpath," + libs[0]]
            subprocess.check_call(cmd)
            return JP2_OUT
    return None


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SRCS + HDRS + ["build.py"])


def build(force=False, extra=(), out=None):
    out = out or OUT
    if out == OUT and not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SRCS:
        tag = "" if out == OUT else "." + os.path.basename(out)[len("liblbdrn_hip_"):-len(".so")]   # objects per variant
        obj = os.path.join(HERE, src.replace(".hip", tag + ".o"))
        objs.append(obj)
        cmd = [hipcc, "-c"] + [f for f in FLAGS if f != "-shared"] + list(extra) + \
              ["-o", obj, os.path.join(HERE, src)]
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(HERE, "exports.map"),
                           "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--stamps" in sys.argv:  # diagnostic build with in-kernel s_memtime stamps (never benchmarked)
        print(build(force=True, extra=["-DLBDRN_TRAIN_STAMPS", "-DLBDRN_APPLY_STAMPS"],
                    out=os.path.join(os.path.dirname(HERE), "liblbdrn_hip_stamps.so")))
    elif "--variant" in sys.argv:  # A/B build: python build.py --variant NAME -DFOO ... -> liblbdrn_hip_NAME.so
        k = sys.argv.index("--variant")
        print(build(force=True, extra=[a for a in sys.argv[k + 2:] if a.startswith("-D")],
                    out=os.path.join(os.path.dirname(HERE), f"liblbdrn_hip_{sys.argv[k + 1]}.so")))
    else:
        print(build(force="--force" in sys.argv))
        print(build_jp2(force="--force" in sys.argv) or "liblbdrn_jp2.so: OpenJPEG not found, not built")
