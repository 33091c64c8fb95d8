"""Builds libmmg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The binary is git-ignored but travels to the GPU box with the tree; `libmmg.so.srchash` beside it records the SHA-256 of
every source it was compiled from (and of the compile flags).  build_library() recompiles whenever that hash differs from
the sources in the tree -- a stale or foreign binary is never used silently -- and check_library() lets a box without a
compiler (or a test) assert that the binary it loads matches the sources it sees."""
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mmg.hip")
OUT = os.path.join(HERE, "libmmg.so")
# (host side at -O1: at -O2 and above X86 instruction selection needs six minutes for the launch glue of this file --
#  the host code only fills argument structs and enqueues launches.  IEEE NaN semantics are kept library-wide; the
#  class-logit inner loops use device_utils.h: fmax_nn instead of building everything with -fno-honor-nans.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-Xarch_host", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.hip"))
                  + [os.path.join(os.path.dirname(HERE), "include", "mmg.h")])


def source_hash(extra_flags=()):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    for path in sources():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


def check_library(out=OUT, extra_flags=()):
    """True when `out` exists and was built from exactly the sources (and flags) in this tree."""
    try:
        return os.path.exists(out) and open(out + ".srchash").read().strip() == source_hash(extra_flags)
    except OSError:
        return False


def build_library(force=False, verbose=True, out=OUT, extra_flags=()):
    if not force and check_library(out, extra_flags):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(out + ".srchash", "w") as f:
        f.write(source_hash(extra_flags) + "\n")
    return out


def build_timing_library(verbose=True):
    """-DMMG_TIMING build (in-kernel s_memrealtime stamps) for scripts/*timeline.py; built on demand, never shipped."""
    extra = tuple(os.environ.get("MMG_EXTRA_FLAGS", "").split())      # (experiments: scripts/ab_*.sh)
    return build_library(out=os.path.join(HERE, "libmmg_timing%s.so" % ("_x" if extra else "")), extra_flags=("-DMMG_TIMING",) + extra, verbose=verbose)


if __name__ == "__main__":
    build_library(force=True)
