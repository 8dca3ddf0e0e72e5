"""Compile oracle/lbdrn_oracle.c and oracle/plane_codec.c into oracle/_build/liblbdrn_oracle.so.

TEST INFRASTRUCTURE.  Called from __graft_entry__.build() and lazily from
oracle/oracle.py.  -ffp-contract=off keeps gcc from fusing the explicit
mul/add pairs, so every rounding in the C text is the rounding that happens.
-mfma is deliberately not passed: fmaf() goes through glibc, which picks the
hardware FMA at run time when the CPU has one, so the .so also runs (slower,
same bits) on a host without FMA.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "lbdrn_oracle.c"), os.path.join(HERE, "plane_codec.c")]
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liblbdrn_oracle.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(map(os.path.getmtime, SRCS)):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math",
           "-Wall", "-o", OUT] + SRCS + ["-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
