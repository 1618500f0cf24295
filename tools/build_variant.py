#!/usr/bin/env python3
"""Developer aid for A/B kernel experiments: build clavaddpm_amd/csrc/variants/libclava_hip_<name>.so with extra -D flags
on the named sources (all other objects are reused from the normal build).

    python tools/build_variant.py prio gemm_f16x3.hip -DCLAVA_GEMM_SETPRIO

tools/*.py pick a variant up through CLAVA_LIB_VARIANT=<name> (see tools/_variant.py); the product never does."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clavaddpm_amd.csrc import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    srcs = [a for a in sys.argv[2:] if a.endswith(".hip")]
    defs = [a for a in sys.argv[2:] if not a.endswith(".hip")]
    B.build(verbose=False)
    out_dir = os.path.join(B.HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in B.sources():
        o = os.path.join(B.OBJ_DIR, src[:-4] + ".o")
        if src in srcs:
            o = os.path.join(out_dir, f"{src[:-4]}_{name}.o")
            subprocess.run([B._hipcc(), *B.FLAGS, *defs, "-c", os.path.join(B.HERE, src), "-o", o], check=True)
        objs.append(o)
    lib = os.path.join(out_dir, f"libclava_hip_{name}.so")
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-fno-gpu-rdc", *objs, "-o", lib], check=True)
    print("built", lib)


if __name__ == "__main__":
    main()
