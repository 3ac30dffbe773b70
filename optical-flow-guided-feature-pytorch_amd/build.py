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
           "pw_tdiff_split.hip", "chain_fused.hip", "chain_split.hip", "winograd.hip", "winograd7.hip", "wino_mid.hip", "wino_gemm.hip", "wino_gemm_split.hip")
HEADERS = ("offk_common.h", "offk_internal.h", "winograd_common.h", os.path.join("..", "..", "include", "offk.h"))
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=fast"]
# heads.hip: no SLP vectoriser.  It pairs the FC accumulators of neighbouring classes into v_pk_fma_f32 with
# op_sel operand swizzles, and on MI355X those produced wrong low-half results whenever the kernel shared
# CUs with another stream's MFMA kernel (measured); scalar FMAs are exact and the
# kernels in that file are latency-bound anyway.
# conv_igemm.hip: its LDS-DMA inline asm writes m0 and says so in the clobber list (the compiler must not assume an m0 value
# of its own survives the statement); clang answers every such statement with "clobber list contains reserved registers".
EXTRA_FLAGS = {"heads.hip": ["-fno-slp-vectorize"], "units_bwd.hip": ["-fno-slp-vectorize"],
               "conv_igemm.hip": ["-Wno-inline-asm"], "pw_tdiff.hip": ["-Wno-inline-asm"], "pw_tdiff_split.hip": ["-Wno-inline-asm", "-fno-slp-vectorize"],
               "chain_fused.hip": ["-Wno-inline-asm"], "wino_gemm_split.hip": ["-fno-slp-vectorize"], "chain_split.hip": ["-fno-slp-vectorize"]}


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


def _code_objects(obj):
    """Unbundles the gfx950 code object(s) of a host object file next to it; returns their paths (the caller removes them)."""
    objdump = _llvm_objdump()
    r = subprocess.run([objdump, "--offloading", obj], capture_output=True, text=True, cwd=os.path.dirname(obj))
    found = []
    for f in sorted(os.listdir(os.path.dirname(obj))):
        if not f.startswith(os.path.basename(obj) + "."):
            continue
        path = os.path.join(os.path.dirname(obj), f)
        if "amdgcn" in f:
            found.append(path)
        elif "host" in f:
            os.remove(path)
    if not found and os.path.basename(obj) != "offk_api.o":       # offk_api.hip has no device code
        raise RuntimeError("no gfx950 code object found in %s (%s)" % (obj, r.stderr.strip()))
    return found


def check_isa(obj, code_objects):
    """The DESIGN.md co-residency finding: compiler-generated packed-fp32 VALU instructions with an op_sel operand
    swizzle (`v_pk_fma_f32 ... op_sel:[0,1,0]`, what the SLP vectoriser makes of neighbouring scalar FMAs that share a
    broadcast operand) returned wrong low halves on MI355X while the kernel shared CUs with an MFMA kernel of another
    stream.  The mechanism is not established, so the hazard is kept out of the library by construction: disassemble
    the gfx950 code object of every translation unit and refuse to build if one contains such an instruction (fix:
    add the file to EXTRA_FLAGS with -fno-slp-vectorize, or break the pairing in the source)."""
    bad = []
    for path in code_objects:
        d = subprocess.run([_llvm_objdump(), "-d", path], capture_output=True, text=True)
        bad += [ln.strip() for ln in d.stdout.splitlines() if _OPSEL_RE.search(ln)]
    if bad:
        raise RuntimeError("ISA check failed for %s: %d packed-fp32 instruction(s) with op_sel (see check_isa), first: %s"
                           % (obj, len(bad), bad[0]))


# ---- resource guard of the asm-scheduled kernels --------------------------------------------------------
# pw_tdiff16_kernel, chain14_kernel and the LDS-DMA form of conv_igemm_kernel (PREC 4) issue their global -> LDS DMAs and some
# operand loads through inline asm and count their own `s_waitcnt vmcnt(N)` ("all but my N newest loads have landed").  The
# memory counter retires in order, so a VMEM instruction hipcc adds on its own -- a register spill, a scratch access -- cannot
# make such a wait too weak; it makes it WAIT FOR THE LOADS JUST ISSUED, i.e. it silently puts a memory latency into the MFMA
# stream the counts were written to keep clear (chain14_kernel<4, true> did exactly that until round 4: four spilled address
# registers, two `s_waitcnt vmcnt(0)` per pass).  And each kernel's block-per-CU plan rests on a register budget.  So the build
# reads the code-object metadata of every kernel named here and fails unless
#   vgpr_spill_count == 0 and private_segment_fixed_size == 0   (no compiler VMEM traffic of its own; SGPR spills are lane
#                                                                 writes into a VGPR -- no memory instruction -- and are
#                                                                 tolerated only where `sgpr_spills` says so, scratch still 0)
#   vgpr_count (unified: arch + acc registers) <= the bound the kernel's occupancy plan assumes.
# (source file, regex on the demangled kernel name, max vgpr_count, SGPR spills tolerated, what the bound stands for)
ASM_SCHEDULED_KERNELS = (
    ("pw_tdiff.hip", r"^offk::pw_tdiff16_kernel\(", 128, True, "four blocks per CU (4 waves / SIMD x 128 = 512; 4 x 29 KB of LDS)"),
    ("pw_tdiff_split.hip", r"^offk::pw_tdiff_split_kernel\(", 256, True,
     "two blocks per CU (2 waves / SIMD; 80 KB of LDS each); its weight loads and feature-map DMAs are asm with hand-counted vmcnt waits"),
    ("chain_fused.hip", r"^void offk::chain14_kernel<", 168, False, "three blocks per CU (52.5 KB of LDS each)"),
    ("wino_gemm.hip", r"^offk::wino_gemm_kernel\(", 128, True, "four blocks per CU (32 KB of LDS each); counts its epilogue's sixteen stores (vmcnt(16))"),
    ("conv_igemm.hip", r"^void offk::conv_igemm_kernel<\d, \d, \d, 1, 1, 2, 2, 4>", 128, False,
     "the default 64x64 tile: four blocks per CU (32 KB of LDS each -- 32768 B is 26 granules of 1280 B, five do not fit; 4 waves / SIMD x 128 = 512)"),
    ("conv_igemm.hip", r"^void offk::conv_igemm_kernel<.*, 4>\(", 512, True, "the other LDS-DMA tiles (set_conv_plan / tools only)"),
)


def _llvm_readelf():
    p = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    return p if os.path.exists(p) else shutil.which("llvm-readelf")


def _cxxfilt():
    """The demangler that ships beside llvm-objdump / llvm-readelf, then whatever PATH has (a bare `c++filt`)."""
    p = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    return p if os.path.exists(p) else (shutil.which("llvm-cxxfilt") or shutil.which("c++filt"))


def kernel_resources(code_object):
    """[{name (demangled), vgpr_count, agpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size}] of one code object."""
    r = subprocess.run([_llvm_readelf(), "--notes", code_object], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("llvm-readelf --notes failed for %s: %s" % (code_object, r.stderr.strip()))
    out = []
    for blk in r.stdout.split("  - .agpr_count:")[1:]:
        def field(k):
            return re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        out.append({"mangled": field("name"), "agpr_count": int(blk.split()[0]), "vgpr_count": int(field("vgpr_count")),
                    "vgpr_spill_count": int(field("vgpr_spill_count")), "sgpr_spill_count": int(field("sgpr_spill_count")),
                    "private_segment_fixed_size": int(field("private_segment_fixed_size"))})
    if out:
        d = subprocess.run([_cxxfilt()] + [k["mangled"] for k in out], capture_output=True, text=True)
        names = d.stdout.splitlines()
        if d.returncode != 0 or len(names) != len(out):
            raise RuntimeError("%s failed on the kernel names of %s: %s" % (_cxxfilt(), code_object, d.stderr.strip()))
        for i, k in enumerate(out):
            k["name"] = names[i]
    return out


# ---- counted waits that leave STORES in flight --------------------------------------------------------------------
# wino_gemm_kernel ends an item with sixteen buffer stores and `s_waitcnt vmcnt(16)`: "everything but my sixteen newest operations has
# completed" is read as "the next tile's LDS-DMA loads have landed".  That rests on (1) vmcnt counting loads and stores of one wave in
# issue order and decrementing in that order (CDNA ISA, "Data dependency resolution": memory reads and writes return in the order
# they were issued for VM_CNT; only loads that return data out of order -- none here: no MUBUF load is in flight behind the stores --
# break it) and (2) hipcc emitting exactly sixteen VMEM instructions, all of them stores, between the last DMA and the wait.  (2) is
# checked here on the disassembly: for every `s_waitcnt vmcnt(N)` of a kernel listed below, the N VMEM instructions in front of it
# must all be buffer stores.
# (source file, regex on the MANGLED symbol, N)
COUNTED_STORE_WAITS = (("wino_gemm.hip", r"wino_gemm_kernel", 16),)
_VMEM_RE = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)\w+")


def check_counted_store_waits(src_name, code_objects):
    rows = []
    for src, pat, n in COUNTED_STORE_WAITS:
        if src != src_name:
            continue
        found = 0
        for path in code_objects:
            d = subprocess.run([_llvm_objdump(), "-d", path], capture_output=True, text=True)
            body, inside = [], False
            for ln in d.stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    inside = re.search(pat, m.group(1)) is not None
                    continue
                if inside:
                    body.append(ln.split("//")[0].strip())
            for i, ins in enumerate(body):
                if not re.match(r"s_waitcnt\b.*vmcnt\(%d\)" % n, ins):
                    continue
                found += 1
                vm = [b for b in body[:i] if _VMEM_RE.match(b)][-n:]
                bad = [b for b in vm if not b.startswith("buffer_store_dword")]
                if len(vm) < n or bad:
                    raise RuntimeError("counted-wait guard failed in %s (%s): the %d VMEM instructions in front of `%s` must all be buffer "
                                       "stores, found %s" % (src_name, pat, n, ins, bad[:3] or "only %d" % len(vm)))
        if not found:
            raise RuntimeError("counted-wait guard: no `s_waitcnt vmcnt(%d)` found in %s of %s (renamed? update COUNTED_STORE_WAITS)" % (n, pat, src_name))
        rows.append((src, pat, n, found))
    return rows


# ---- counted waits of the split units kernel --------------------------------------------------------------------------
# pw_tdiff_split_kernel orders its asm loads (weights -> registers, feature map -> LDS by DMA) with three counted waits per K-tile step;
# each count rests on the step issuing EXACTLY this sequence of vector-memory instructions and nothing else (a compiler-added VMEM
# instruction -- spill, scratch -- would make a wait cover the wrong operations):
#     s_waitcnt vmcnt(7) | 6 register loads (Wg), 3 LDS-DMA (pieces 0-2) | s_waitcnt vmcnt(9) | 1 LDS-DMA (piece 3), 3 register loads (Wd) | s_barrier
# Checked on the disassembly for every step body (the loop is unrolled by two + a tail step); and no v_mov / v_accvgpr may read a
# register an in-flight load of the step writes before the wait that covers it (a copy of a load destination = stale operands).
# (source file, regex on the mangled symbol, first wait, ops, second wait, ops)
COUNTED_LOAD_STEPS = (("pw_tdiff_split.hip", r"pw_tdiff_split_kernel", 7, ("r",) * 6 + ("d",) * 3, 9, ("d",) + ("r",) * 3),)
_LOAD_RE = re.compile(r"^buffer_load_dwordx4\s+(?:v\[(\d+):(\d+)\]|v(\d+)),.*?(\blds\b)?$")


def counted_load_steps(body, w1, ops1, w2, ops2):
    """body: instructions of one kernel.  Returns the number of step bodies found; raises on a malformed one."""
    steps = 0
    i = 0
    while i < len(body):
        if not re.match(r"s_waitcnt\b.*vmcnt\(%d\)" % w1, body[i]):
            i += 1
            continue
        j, seq, inflight = i + 1, [], []
        phase, ok = 0, False
        while j < len(body):
            ins = body[j]
            if ins.startswith("s_barrier"):
                ok = phase == 1
                break
            if re.match(r"s_waitcnt\b.*vmcnt\(%d\)" % w2, ins) and phase == 0:
                if tuple(seq) != tuple(ops1):
                    raise RuntimeError("between vmcnt(%d) and vmcnt(%d): VMEM sequence %s, expected %s" % (w1, w2, "".join(seq), "".join(ops1)))
                phase, seq = 1, []
            elif re.match(r"s_waitcnt\b.*vmcnt\(", ins):
                break                                           # some other wait: not a step body (prologue / epilogue)
            elif _VMEM_RE.match(ins):
                m = _LOAD_RE.match(ins)
                if not m:
                    raise RuntimeError("unexpected vector-memory instruction inside a step: %s" % ins)
                if ins.rstrip().endswith("lds"):
                    seq.append("d")
                else:
                    seq.append("r")
                    inflight.append((int(m.group(1)), int(m.group(2))))
            elif re.match(r"v_(mov_b32|accvgpr_)", ins):
                srcs = [int(x) for x in re.findall(r"\bv(\d+)\b", ins.split(",", 1)[1] if "," in ins else "")]
                for lo, hi in inflight:
                    if any(lo <= r <= hi for r in srcs):
                        raise RuntimeError("copy of an in-flight load destination inside a step: %s" % ins)
            j += 1
        if ok:
            if tuple(seq) != tuple(ops2):
                raise RuntimeError("between vmcnt(%d) and the step's barrier: VMEM sequence %s, expected %s" % (w2, "".join(seq), "".join(ops2)))
            steps += 1
        i = j
    return steps


def check_counted_load_steps(src_name, code_objects):
    for src, pat, w1, ops1, w2, ops2 in COUNTED_LOAD_STEPS:
        if src != src_name:
            continue
        found = 0
        for path in code_objects:
            d = subprocess.run([_llvm_objdump(), "-d", path], capture_output=True, text=True)
            body, inside = [], False
            for ln in d.stdout.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    inside = re.search(pat, m.group(1)) is not None
                    continue
                if inside and ln.strip():
                    body.append(ln.split("//")[0].strip())
            try:
                found += counted_load_steps(body, w1, ops1, w2, ops2)
            except RuntimeError as e:
                raise RuntimeError("counted-load guard failed in %s (%s): %s" % (src_name, pat, e))
        if found < 2:
            raise RuntimeError("counted-load guard: %d step bodies found in %s of %s, expected >= 2 (renamed? update COUNTED_LOAD_STEPS)" % (found, pat, src_name))


# A VALU write to a data register of a 12- / 16-byte store in the instruction directly behind the store: on MI355X the new value reached
# memory in some lanes (wino_gemm_split.hip, round 5: lanes 12-15 of every sixteen, run-to-run varying).  LLVM's hazard recogniser
# places the wait state only when the store has no SGPR offset; hipcc had re-used the first data register for the next address.
# (the DATA registers: the first operand of a buffer store, the second of a global / flat store -- whose first is the address pair)
_WIDE_STORE_RE = re.compile(r"^(?:buffer_store_dwordx[34]\s+|(?:global|flat)_store_dwordx[34]\s+v\[\d+:\d+\],\s*)v\[(\d+):(\d+)\]")
_VALU_DST_RE = re.compile(r"^v_(?!cmp|cmpx|readlane|readfirstlane)\w+\s+(?:v\[(\d+):(\d+)\]|v(\d+))(?=[,\s]|$)")


def store_data_hazards(disassembly):
    """A wide store followed -- directly, or with ONE non-VALU instruction (an SALU op, an s_nop 0) in between -- by a VALU write to one
    of its data registers.  profiles/r05/probe_store_hazard.txt: with an SGPR soffset one wait state cures it, with a constant soffset
    (the case LLVM's recogniser handles itself) two are needed; the scan covers both distances so that neither rests on the recogniser.
    `s_nop N` with N >= 1 between the two ends the window (that is the cure)."""
    body = [ln.split("//")[0].strip() for ln in disassembly.splitlines()]
    body = [b for b in body if b and not b.endswith(":")]
    bad = []
    for i, a in enumerate(body):
        m = _WIDE_STORE_RE.match(a)
        if not m:
            continue
        for dist in (1, 2):
            if i + dist >= len(body):
                break
            b = body[i + dist]
            nop = re.match(r"s_nop\s+(\d+)", b)
            if nop and int(nop.group(1)) + 1 >= 3 - dist:      # wait states left to cover: 2 at distance 1, 1 at distance 2
                break
            d = _VALU_DST_RE.match(b)
            if d:
                lo, hi = (int(d.group(3)),) * 2 if d.group(3) else (int(d.group(1)), int(d.group(2)))
                if lo <= int(m.group(2)) and hi >= int(m.group(1)):
                    bad.append("%s  ||  %s" % (a, b))
                break                                          # a VALU instruction fills the window whatever it writes... only the first one is looked at
            if b.startswith("v_") or _WIDE_STORE_RE.match(b):
                break
    return bad


def check_store_data_hazard(obj, code_objects):
    bad = []
    for path in code_objects:
        bad += store_data_hazards(subprocess.run([_llvm_objdump(), "-d", path], capture_output=True, text=True).stdout)
    if bad:
        raise RuntimeError("ISA check failed for %s: %d wide store(s) with a VALU write to their data registers directly behind them "
                           "(put `s_nop 1` between fences behind the store, see wino_gemm_split.hip), first: %s" % (obj, len(bad), bad[0]))


def check_resources(src_name, kernels):
    """Applies ASM_SCHEDULED_KERNELS to the kernels of one translation unit; returns the rows it checked."""
    rows, matched = [], set()
    for k in kernels:
        for gi, (src, pat, max_vgpr, sgpr_ok, why) in enumerate(ASM_SCHEDULED_KERNELS):
            if src != src_name or not re.search(pat, k["name"]):
                continue
            matched.add(gi)
            problems = []
            if k["vgpr_spill_count"] or k["private_segment_fixed_size"]:
                problems.append("vgpr_spill_count %d, private_segment_fixed_size %d (must be 0: its vmcnt waits are hand-counted)"
                                % (k["vgpr_spill_count"], k["private_segment_fixed_size"]))
            if k["sgpr_spill_count"] and not sgpr_ok:
                problems.append("sgpr_spill_count %d (must be 0)" % k["sgpr_spill_count"])
            if k["vgpr_count"] > max_vgpr:
                problems.append("vgpr_count %d > %d: %s" % (k["vgpr_count"], max_vgpr, why))
            if problems:
                raise RuntimeError("resource guard failed for %s in %s: %s" % (k["name"], src_name, "; ".join(problems)))
            rows.append(k)
            break
    for gi, (src, pat, _m, _s, _w) in enumerate(ASM_SCHEDULED_KERNELS):
        if src == src_name and gi not in matched:
            raise RuntimeError("resource guard: no kernel of %s matches %r (renamed? update ASM_SCHEDULED_KERNELS)" % (src_name, pat))
    return rows


def check_object(obj):
    """ISA check + resource guard of one freshly compiled object."""
    if _llvm_objdump() is None or _llvm_readelf() is None or _cxxfilt() is None:
        if os.environ.get("OFFK_SKIP_ISA_CHECK") == "1":
            print("WARNING: llvm-objdump / llvm-readelf / llvm-cxxfilt not found, checks of %s SKIPPED (OFFK_SKIP_ISA_CHECK=1)" % obj, file=sys.stderr)
            return
        raise RuntimeError("llvm-objdump / llvm-readelf / llvm-cxxfilt (or c++filt) not found: cannot run the op_sel ISA check and the "
                           "resource guard; OFFK_SKIP_ISA_CHECK=1 builds without them")
    cos = _code_objects(obj)
    try:
        check_isa(obj, cos)
        check_store_data_hazard(obj, cos)
        src = os.path.basename(obj).replace(".o", ".hip")
        kernels = [k for co in cos for k in kernel_resources(co)]
        check_resources(src, kernels)
        check_counted_store_waits(src, cos)
        check_counted_load_steps(src, cos)
    finally:
        for path in cos:
            os.remove(path)


def resource_table():
    """The guarded kernels of the objects of the last build, for tests/test_build_guard.py: {source: [rows]}."""
    table = {}
    for src in sorted(set(g[0] for g in ASM_SCHEDULED_KERNELS)):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cos = _code_objects(obj)
        try:
            table[src] = check_resources(src, [k for co in cos for k in kernel_resources(co)])
        finally:
            for path in cos:
                os.remove(path)
    return table


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
        try:
            check_object(o)
        except Exception:
            os.remove(o)          # a refused object must not pass for a current one on the next build
            raise
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
