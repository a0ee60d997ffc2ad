"""Builds libquantool_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree output: ``quantool_amd/lib/libquantool_hip.so`` (git-ignored, travels to the GPU box).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
LIB_DIR = PKG / "lib"
OBJ_DIR = CSRC / "_obj"
LIB_PATH = LIB_DIR / "libquantool_hip.so"

# -ffp-contract=off: the sweep / prepare kernels restate upstream's op-by-op fp32 arithmetic;
# contraction to FMA would change roundings.  MFMA builtins are unaffected.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-fPIC",
    "-std=c++17",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
    # per-kernel register / scratch / LDS figures as compiler remarks: parsed by _resources() below, not printed
    "-Rpass-analysis=kernel-resource-usage",
]

# Kernels allowed to use scratch (private memory): none.  Round 4 found every sgemm_tn_kernel variant holding its 264-byte
# argument struct in scratch -- the kernel modified the by-value struct and indexed an array in it at run time, so
# every pointer and pitch inside the k-loop became a scratch load; the chains' small products ran 10-25 % longer and
# nothing failed.  Now any kernel with a private segment fails the build.
SCRATCH_OK: tuple = ()

# lab builds: extra flags (e.g. QT_EXTRA_HIPCC_FLAGS=-DQT_XTX_ABLATION for tools/xtx_wrap_sweep.sh); part of the stamp
HIPCC_FLAGS += os.environ.get("QT_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stamp(src: Path) -> str:
    h = hashlib.sha1()
    h.update(src.read_bytes())
    for hdr in sorted(list(CSRC.glob("*.h")) + [PKG.parent / "include" / "quantool_amd.h"]):
        h.update(hdr.read_bytes())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _compile(src: Path) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    stamp = OBJ_DIR / (src.stem + ".stamp")
    want = _stamp(src)
    if obj.exists() and stamp.exists() and stamp.read_text() == want:
        return obj
    cmd = [_hipcc(), *HIPCC_FLAGS, "-c", str(src), "-o", str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    rest = _resources(src, res.stderr)
    if rest.strip():
        sys.stderr.write(rest)
    stamp.write_text(want)
    return obj


def _resources(src: Path, stderr: str) -> str:
    """Split the kernel-resource-usage remarks out of hipcc's stderr: one line per kernel goes to
    ``_obj/<stem>.resources.txt`` (name, VGPRs, AGPRs, scratch bytes per lane, spills, LDS bytes, waves per SIMD), a kernel
    with scratch that is not in SCRATCH_OK fails the build.  Returns what is left of stderr (real warnings)."""
    import re

    kernels, cur, rest = [], None, []
    lines = stderr.splitlines()
    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if m:
            kv = m.group(1).strip()
            if kv.startswith("Function Name:"):
                cur = {"name": kv.split(":", 1)[1].strip()}
                kernels.append(cur)
            elif cur is not None and ":" in kv:
                k, v = kv.rsplit(":", 1)
                cur[k.strip()] = v.strip()
            # the remark is followed by a source excerpt and a caret line on its first occurrence per location
            while i + 1 < len(lines) and re.match(r"^\s+\d* ?\|", lines[i + 1]):
                i += 1
        elif not re.match(r"^\d+ (warning|remark)s? generated", line.strip()) and "remarks generated" not in line:
            rest.append(line)
        i += 1
    rows = []
    for k in kernels:
        rows.append(f"{k['name']}\tvgpr {k.get('VGPRs', '?')}\tagpr {k.get('AGPRs', '?')}\tscratch "
                    f"{k.get('ScratchSize [bytes/lane]', '?')}\tvgpr_spill {k.get('VGPRs Spill', '?')}\tlds "
                    f"{k.get('LDS Size [bytes/block]', '?')}\twaves_per_simd {k.get('Occupancy [waves/SIMD]', '?')}")
    (OBJ_DIR / (src.stem + ".resources.txt")).write_text("\n".join(rows) + ("\n" if rows else ""))
    bad = [k["name"] for k in kernels
           if k.get("ScratchSize [bytes/lane]", "0") not in ("0", "?") and not any(ok in k["name"] for ok in SCRATCH_OK)]
    if bad:
        raise RuntimeError(f"{src.name}: kernel(s) with a private segment (scratch): {bad} -- see "
                           f"{OBJ_DIR / (src.stem + '.resources.txt')}; a by-value argument struct that is modified and "
                           "indexed at run time, or a register array indexed at run time, is the usual cause")
    return "\n".join(rest) + ("\n" if rest else "")


def audit_m0(src: Path) -> None:
    """xtx.hip, gemm3_tn.hip and sgemm_tn.hip write M0 from inline asm without restoring it (the LDS-DMA destination).  That is only
    sound while hipcc itself never touches M0 in that translation unit, so the device ISA is checked:
    every line that names m0 must sit inside an ;;#ASMSTART ... ;;#ASMEND block."""
    stamp = OBJ_DIR / (src.stem + ".m0audit")
    want = _stamp(src)
    if stamp.exists() and stamp.read_text() == want:
        return
    asm = OBJ_DIR / (src.stem + ".device.s")
    cmd = [_hipcc(), *[f for f in HIPCC_FLAGS if f != "-fPIC"], "--cuda-device-only", "-S", str(src), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc -S failed for {src.name}:\n{res.stderr}")
    inside = False
    for n, line in enumerate(asm.read_text().splitlines(), 1):
        if ";;#ASMSTART" in line:
            inside = True
        elif ";;#ASMEND" in line:
            inside = False
        elif not inside and "m0" in line.split(";")[0].replace("_m0", ""):
            raise RuntimeError(f"{src.name}: compiler-generated use of M0 at {asm.name}:{n}: {line.strip()!r} -- "
                               "the LDS-DMA helpers must save/restore M0 again")
    audit_spills(src.name, asm.read_text())
    asm.unlink()
    stamp.write_text(want)


# kernels whose inner loops are hand-scheduled around a fixed register budget: a spill there is a silent 20-30 %
# (round 3: wrapping gemm3_kernel's body in an item loop let LICM hoist 128 epilogue addresses -> 179 spills)
NO_SPILL_KERNELS = ("xtx_kernel", "xtx16_kernel", "gemm3_kernel", "sgemm_ring_kernel")


def audit_spills(src_name: str, asm_text: str) -> None:
    """Fail the build when one of NO_SPILL_KERNELS uses scratch (the kernel metadata hipcc -S prints)."""
    name = None
    for line in asm_text.splitlines():
        t = line.strip()
        if t.startswith(".name:"):
            name = t.split(":", 1)[1].strip()
        elif name and t.startswith((".vgpr_spill_count:", ".private_segment_fixed_size:")):
            if int(t.split(":", 1)[1]) != 0 and any(k in name for k in NO_SPILL_KERNELS):
                raise RuntimeError(f"{src_name}: {name} spills ({t}) -- its schedule assumes every value stays in registers")


def build(verbose: bool = False) -> Path:
    OBJ_DIR.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    for name in ("xtx.hip", "gemm3_tn.hip", "sgemm_tn.hip"):
        audit_m0(CSRC / name)
    newest = max(o.stat().st_mtime for o in objs)
    if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(verbose=True)
