#!/usr/bin/env python3
"""Build the drop-in CPython extension modules in-tree (gcc; links libpysparse_hip.so).

  pysparse_amd/sparse/spmatrix.*.so      (pysparse.sparse.spmatrix)
  pysparse_amd/itsolvers/krylov.*.so     (pysparse.itsolvers.krylov)
  pysparse_amd/precon/precon.*.so        (pysparse.precon.precon)
"""
import os
import subprocess
import sys
import sysconfig

import numpy

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
MODULES = [
    ("sparse", "spmatrix", "sparse/src/spmatrixmodule.c"),
    ("itsolvers", "krylov", "itsolvers/src/krylovmodule.c"),
    ("precon", "precon", "precon/src/preconmodule.c"),
]
HEADERS = ["include/spmatrix_api.h", "include/psp_pyops.h"]


def build(force=False):
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    cc = os.environ.get("CC", "gcc")
    incs = ["-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(),
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "include")]
    lib = os.path.join(PKG, "libpysparse_hip.so")
    if not os.path.exists(lib):
        raise SystemExit("build libpysparse_hip.so first (__graft_entry__.build_hip_library)")
    for sub, name, src in MODULES:
        out = os.path.join(PKG, sub, name + suffix)
        srcp = os.path.join(PKG, src)
        deps = [srcp, lib, os.path.join(ROOT, "include", "pysparse_hip.h")] + [os.path.join(PKG, h) for h in HEADERS]
        if name == "spmatrix":
            deps.append(os.path.join(PKG, "sparse/src/ll_mat_edit.c"))  # textually included by spmatrixmodule.c
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
            continue
        # PSP_EXT_CFLAGS: extra flags, e.g. "-fsanitize=address,undefined -g -O1" for tools/sanitize_host.sh
        cmd = [cc, "-O2", "-fPIC", "-shared", "-pthread", "-std=gnu99", "-Wall", "-Wno-unused-function"] + \
              os.environ.get("PSP_EXT_CFLAGS", "").split() + incs + \
              [srcp, "-o", out, "-L" + PKG, "-lpysparse_hip", "-Wl,-rpath,$ORIGIN/.."]
        subprocess.check_call(cmd)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
