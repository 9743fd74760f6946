"""Builds the C-ABI libraries of include/btr_pointnet2.h with hipcc for gfx950.

In-tree build: the .so files land in backtoreality_amd/lib/ so that they travel with the repo
snapshot to the GPU box (git-ignored, not gpurun-ignored).  hipcc cross-compiles without a
GPU, so this also runs in the CPU-only build container.

One library per rounding mode of the squared distance (BTR_FMAD, csrc/common.hpp):
    libbtr_pointnet2.so         BTR_FMAD=1  (default: what an nvcc --fmad=true build rounds to)
    libbtr_pointnet2_fmad0.so   BTR_FMAD=0  (as written, no contraction)
    libbtr_pointnet2_fmad2.so   BTR_FMAD=2  (left-to-right fma chain)
Only the index-producing sources depend on the mode; the other objects are shared.  Every
source is compiled to its own object (in parallel, cached by content hash) and linked.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_NAMES = {1: "libbtr_pointnet2.so", 0: "libbtr_pointnet2_fmad0.so",
             2: "libbtr_pointnet2_fmad2.so"}
LIB_PATH = os.path.join(LIB_DIR, LIB_NAMES[1])

ARCH = "gfx950"
# -ffp-contract=off: index-producing kernels must round f32 expressions exactly as the mode
# says (bit-exact parity with the oracle); nothing may fuse on its own.  Files that want FMA
# contraction (MFMA GEMM tiles) opt back in with `#pragma clang fp contract(fast)`.
HIPCC_FLAGS = [
    "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
    "-fPIC", "-Wall", "-Wno-unused-function",
    # No SLP vectorisation: it turns neighbouring scalar f32 ops into packed v_pk_{add,mul,fma}_f32
    # with op_sel operands.  With such code the register-resident FPS kernel returned a wrong
    # sequence in 1-3 % of its launches whenever another stream's kernels shared the chip
    # (never alone; identical inputs, tools/diag_pipeline_inds.py) and not once in 10 000
    # launches without it; the guide lists packed f32 as an anti-lever beside MFMAs anyway.
    "-fno-slp-vectorize",
    # ... and no loop vectorisation either (round 6: the whole library, not only the index-producing
    # sources): it packed f32 ops in the loss kernels, the LayerNorm backward and the attention
    # kernels too (929 sites), and the guard below had nothing to say about them.  Memory
    # operations are merged by another pass and are unaffected.
    "-fno-vectorize",
]
# sources whose code depends on BTR_FMAD (they evaluate sq3() / dot3()): the index-producing ones
MODE_SOURCES = ("ball_query.hip", "ball_query_bucket.hip", "ball_query_grid.hip",
                "fps_bucket.hip", "interpolate.hip", "sampling.hip")
# per-file additions to HIPCC_FLAGS (none at present)
EXTRA_FLAGS = {}
# Build-time guard (round-3 review, What's weak #3; round 6: EVERY object): the device code of
# every object is disassembled and the build FAILS on any packed f32 arithmetic instruction --
# the form the wrong-FPS-sequence-under-concurrency incident was bisected to (DESIGN.md 7.5;
# stand-alone reproducer: tools/probe/pk_hazard.hip).  A new compiler or an edit that brings
# them back is caught here, not by a 1-3 % statistical test on the GPU -- and for the float
# kernels, whose results nobody compares bit for bit, not at all otherwise.
PACKED_F32 = r"\bv_pk_(add|mul|fma)_f32\b"
LLVM_BIN = os.environ.get("BTR_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def lib_path(mode=1):
    return os.path.join(LIB_DIR, LIB_NAMES[int(mode)])


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_digest():
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    files.append(os.path.join(os.path.dirname(PKG_DIR), "include", "btr_pointnet2.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h


def _object_plan():
    """[(source, mode or None, object path)]: one object per source, three for MODE_SOURCES."""
    base = _headers_digest()
    plan = []
    for src in sources():
        name = os.path.basename(src)
        with open(src, "rb") as fh:
            body = fh.read()
        for mode in ((0, 1, 2) if name in MODE_SOURCES else (None,)):
            h = base.copy()
            h.update(body)
            h.update(str(mode).encode())
            tag = "" if mode is None else ".fmad%d" % mode
            plan.append((src, mode, os.path.join(
                OBJ_DIR, "%s%s.%s.o" % (name[:-4], tag, h.hexdigest()[:16]))))
    return plan


def build_id(plan=None):
    """Digest of every source, header and flag the libraries are built from: what
    btr_build_id() returns.  profiles/pmc_traffic.json records it, so that bench.py can tell
    whether the counters it quotes were measured on the kernels it is running."""
    plan = plan or _object_plan()
    h = hashlib.sha256()
    for _, _, obj in plan:
        h.update(os.path.basename(obj).encode())
    return h.hexdigest()[:16]


def _build_id_object(plan):
    return os.path.join(OBJ_DIR, "build_id.%s.o" % build_id(plan))


def disassemble(obj):
    """Text disassembly of the gfx950 code object bundled in a hipcc host object."""
    import shutil
    import tempfile
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    with tempfile.TemporaryDirectory(prefix="btr_dis_") as tmp:
        local = os.path.join(tmp, "x.o")
        shutil.copy(obj, local)
        subprocess.check_call([objdump, "--offloading", local], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not dev:
            return ""   # a source without device code (graph_cache.hip: host logic only)
        if len(dev) != 1:
            raise RuntimeError("no single gfx950 bundle in %s: %r" % (obj, dev))
        return subprocess.check_output([objdump, "-d", os.path.join(tmp, dev[0])]).decode()


def packed_f32_sites(obj):
    """[(kernel symbol, instruction line)] of packed f32 arithmetic in an object's device code."""
    import re
    pat, sites, fn = re.compile(PACKED_F32), [], "?"
    for line in disassemble(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            fn = m.group(1)
        elif pat.search(line):
            sites.append((fn, line.strip()))
    return sites


def _guard_marker(obj):
    return obj[:-2] + ".nopk"


def check_objects(plan=None, force=False):
    """Fails (RuntimeError) when an object holds packed f32 arithmetic; the verdict per object is
    remembered beside it (objects are content-addressed)."""
    plan = plan or _object_plan()
    for src, mode, obj in plan:
        if os.path.exists(_guard_marker(obj)) and not force:
            continue
        sites = packed_f32_sites(obj)
        if sites:
            raise RuntimeError(
                "packed f32 arithmetic in an object of the library (%s, BTR_FMAD=%s):\n  %s"
                % (os.path.basename(src), mode,
                   "\n  ".join("%s: %s" % s for s in sites[:12])))
        with open(_guard_marker(obj), "w") as fh:
            fh.write("no v_pk_{add,mul,fma}_f32\n")


def is_fresh():
    plan = _object_plan()
    if not all(os.path.exists(o) for _, _, o in plan) or \
            not os.path.exists(_build_id_object(plan)) or \
            not all(os.path.exists(_guard_marker(o)) for _, m, o in plan):
        return False
    newest = max(os.path.getmtime(o) for _, _, o in plan)
    return all(os.path.exists(lib_path(m)) and os.path.getmtime(lib_path(m)) >= newest
               for m in LIB_NAMES)


def build(force=False, verbose=False, jobs=None):
    """Compile every HIP source and link the three libraries; no-op when nothing changed."""
    if not force and is_fresh():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    plan = _object_plan()
    todo = [(s, m, o) for s, m, o in plan if force or not os.path.exists(o)]

    def compile_one(item):
        src, mode, obj = item
        cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + \
            ["-c", src, "-o", obj + ".tmp"]
        if mode is not None:
            cmd.insert(1, "-DBTR_FMAD=%d" % mode)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)

    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(compile_one, todo))
    check_objects(plan)
    bid_obj = _build_id_object(plan)
    if not os.path.exists(bid_obj):   # (host-only: a second to compile)
        src = bid_obj[:-2] + ".cpp"
        with open(src, "w") as fh:
            fh.write('extern "C" const char *btr_build_id(void) { return "%s"; }\n' % build_id(plan))
        subprocess.check_call([hipcc, "-O1", "-fPIC", "-c", src, "-o", bid_obj + ".tmp"])
        os.replace(bid_obj + ".tmp", bid_obj)
        os.remove(src)
    keep = {o for _, _, o in plan} | {bid_obj} | \
        {_guard_marker(o) for _, m, o in plan}
    for f in os.listdir(OBJ_DIR):                      # objects of older source versions
        if os.path.join(OBJ_DIR, f) not in keep:
            os.remove(os.path.join(OBJ_DIR, f))
    for mode in LIB_NAMES:
        objs = [o for _, m, o in plan if m is None or m == mode] + [bid_obj]
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", lib_path(mode)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))


check_index_objects = check_objects   # (the name rounds 3 - 5 knew the guard by)
