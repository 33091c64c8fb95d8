"""Per-kernel resource usage of the gfx950 device code (registers, spills, scratch, LDS, occupancy) and instruction counts.

  python scripts/isa_stats.py [--kernel SUBSTR] [--count MNEMONIC ...] [-D MACRO ...]

Compiles csrc/mmg.hip with `--cuda-device-only -S` (same flags as multimodalgame_amd/build.py) and parses the per-kernel
trailers the AMDGPU backend writes into the assembly ("; NumVgprs: ...", "; ScratchSize: ...")."""
import argparse
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimodalgame_amd import build  # noqa: E402


def device_asm(out, defines=()):
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-D" + d for d in defines] + ["--cuda-device-only", "-S", "-o", out, build.SRC]
    subprocess.check_call(cmd)
    return open(out).read()


def parse(asm, count=()):
    """{kernel: {field: value, mnemonic: count}}; a kernel's text runs from its label to its `.end_amdhsa_kernel` trailer."""
    kernels = {}
    parts = re.split(r"^(_Z[\w$.]+):\s*(?:;.*)?$", asm, flags=re.M)
    for i in range(1, len(parts) - 1, 2):
        name, body = parts[i], parts[i + 1]
        m = re.search(r"; NumVgprs: (\d+)", body)
        if not m:
            continue
        info = {}
        for key, pat in (("vgpr", r"; NumVgprs: (\d+)"), ("agpr", r"; NumAgprs: (\d+)"), ("sgpr", r"; NumSgprs: (\d+)"),
                         ("scratch", r"; ScratchSize: (\d+)"), ("lds", r"; LDSByteSize: (\d+)"), ("occupancy", r"; Occupancy: (\d+)"),
                         ("sgpr_spill", r"; SGPRSpill: (\d+)|; NumSGPRsForWavesPerEU: (\d+)"), ("vgpr_spill", r"; VGPRSpill: (\d+)")):
            mm = re.search(pat, body)
            info[key] = int(next(g for g in mm.groups() if g is not None)) if mm else None
        code = body.split(".section")[0]
        info["instructions"] = len(re.findall(r"^\s+[vsgdb][_a-z0-9]+ ", code, flags=re.M))
        for mn in ("scratch_store", "scratch_load", "v_readlane", "v_writelane", "v_mfma") + tuple(count):
            info[mn] = len(re.findall(r"^\s+" + re.escape(mn), code, flags=re.M))
        kernels[name] = info
    return kernels


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + list(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="")
    ap.add_argument("--count", nargs="*", default=[])
    ap.add_argument("-D", dest="defines", action="append", default=[])
    ap.add_argument("--asm", default="/tmp/mmg_device.s")
    ap.add_argument("--reuse", action="store_true", help="parse an existing --asm file instead of recompiling")
    a = ap.parse_args()
    asm = open(a.asm).read() if a.reuse else device_asm(a.asm, a.defines)
    ks = parse(asm, a.count)
    dm = demangle(list(ks))
    for n, info in sorted(ks.items(), key=lambda kv: dm[kv[0]]):
        short = re.sub(r"\(.*", "", dm[n]).replace("void mmg::", "")
        if a.kernel and a.kernel not in short:
            continue
        print("%-62s vgpr %3s agpr %3s sgpr %3s  scratch %5s B  vgpr_spill %4s  lds %6s  occ %s  instr %6d  mfma %5d  scratch_ld/st %d/%d  readlane/writelane %d/%d %s" % (
            short[:62], info["vgpr"], info["agpr"], info["sgpr"], info["scratch"], info["vgpr_spill"], info["lds"], info["occupancy"],
            info["instructions"], info["v_mfma"], info["scratch_load"], info["scratch_store"], info["v_readlane"], info["v_writelane"],
            " ".join("%s %d" % (m, info[m]) for m in a.count)))


if __name__ == "__main__":
    main()
