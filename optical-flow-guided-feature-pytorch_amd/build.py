"""Builds liboffk.so (HIP, gfx950 only) in-tree with hipcc.

    python optical-flow-guided-feature-pytorch_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box
with the working tree.
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "liboffk.so")
SOURCES = ("offk_api.hip", "pw_reduce.hip", "sobel_tdiff.hip", "conv_igemm.hip", "heads.hip", "units_bwd.hip", "pw_tdiff.hip",
           "chain_fused.hip", "winograd.hip", "winograd7.hip")
HEADERS = ("offk_common.h", "offk_internal.h", os.path.join("..", "..", "include", "offk.h"))
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=fast"]
# heads.hip: no SLP vectoriser.  It pairs the FC accumulators of neighbouring classes into v_pk_fma_f32 with
# op_sel operand swizzles, and on MI355X those produced wrong low-half results whenever the kernel shared
# CUs with another stream's MFMA kernel (measured); scalar FMAs are exact and the
# kernels in that file are latency-bound anyway.
# conv_igemm.hip: its LDS-DMA inline asm writes m0 and says so in the clobber list (the compiler must not assume an m0 value
# of its own survives the statement); clang answers every such statement with "clobber list contains reserved registers".
EXTRA_FLAGS = {"heads.hip": ["-fno-slp-vectorize"], "units_bwd.hip": ["-fno-slp-vectorize"],
               "conv_igemm.hip": ["-Wno-inline-asm"], "pw_tdiff.hip": ["-Wno-inline-asm"],
               "chain_fused.hip": ["-Wno-inline-asm"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _llvm_objdump():
    if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        return "/opt/rocm/lib/llvm/bin/llvm-objdump"
    return shutil.which("llvm-objdump")


# v_pk_{fma,mul,add}_f32 with a NON-DEFAULT op_sel operand (the implicated form: op_sel:[0,1,0] = low half of the result
# reads the HIGH half of a source).  op_sel_hi alone is not it: the default op_sel_hi is all ones
# and a harmless instruction may print a non-default one.
_OPSEL_RE = re.compile(r"\bv_pk_\w+_f32\b.*\bop_sel:\[")


def check_isa(obj):
    """The DESIGN.md co-residency finding: compiler-generated packed-fp32 VALU instructions with an op_sel operand
    swizzle (`v_pk_fma_f32 ... op_sel:[0,1,0]`, what the SLP vectoriser makes of neighbouring scalar FMAs that share a
    broadcast operand) returned wrong low halves on MI355X while the kernel shared CUs with an MFMA kernel of another
    stream.  The mechanism is not established, so the hazard is kept out of the library by construction: disassemble
    the gfx950 code object of every translation unit and refuse to build if one contains such an instruction (fix:
    add the file to EXTRA_FLAGS with -fno-slp-vectorize, or break the pairing in the source)."""
    objdump = _llvm_objdump()
    if objdump is None:
        if os.environ.get("OFFK_SKIP_ISA_CHECK") == "1":
            print("WARNING: llvm-objdump not found, op_sel ISA check of %s SKIPPED (OFFK_SKIP_ISA_CHECK=1)" % obj, file=sys.stderr)
            return
        raise RuntimeError("llvm-objdump not found: cannot run the op_sel ISA check (DESIGN.md section 8); "
                           "OFFK_SKIP_ISA_CHECK=1 builds without it")
    r = subprocess.run([objdump, "--offloading", obj], capture_output=True, text=True, cwd=os.path.dirname(obj))
    bad = []
    found = False
    for f in sorted(os.listdir(os.path.dirname(obj))):
        if f.startswith(os.path.basename(obj) + ".") and "amdgcn" in f:
            found = True
            path = os.path.join(os.path.dirname(obj), f)
            d = subprocess.run([objdump, "-d", path], capture_output=True, text=True)
            bad += [ln.strip() for ln in d.stdout.splitlines() if _OPSEL_RE.search(ln)]
            os.remove(path)
        elif f.startswith(os.path.basename(obj) + ".") and "host" in f:
            os.remove(os.path.join(os.path.dirname(obj), f))
    if not found and os.path.basename(obj) != "offk_api.o":       # offk_api.hip has no device code
        raise RuntimeError("ISA check: no gfx950 code object found in %s (%s)" % (obj, r.stderr.strip()))
    if bad:
        raise RuntimeError("ISA check failed for %s: %d packed-fp32 instruction(s) with op_sel (see check_isa), first: %s"
                           % (obj, len(bad), bad[0]))


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or not _newer(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc, "-c", s, "-o", o] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        check_isa(o)
        return o

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or force or not _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
