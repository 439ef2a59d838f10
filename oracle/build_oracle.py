"""TEST INFRASTRUCTURE — builds oracle/rasterize_oracle.c into oracle/liboracle_raster.so.

gcc only, no -march (x86-64 baseline: no FMA), -ffp-contract=off: the oracle must round
exactly like the reference's CPU build does.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rasterize_oracle.c")
HDR = os.path.join(HERE, "rasterize_oracle_impl.h")
OUT = os.path.join(HERE, "liboracle_raster.so")


def build(force=False):
    newest = max(os.path.getmtime(SRC), os.path.getmtime(HDR))
    if not force and os.path.isfile(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-std=c99",
           "-o", OUT, SRC, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
