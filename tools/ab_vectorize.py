#!/usr/bin/env python3
"""Same-box A/B of `-fno-vectorize` for the float kernels (build.py, round 6): the shipped
library against one whose `attention.hip`, `decoder.hip`, `gf_loss.hip` and `votenet_loss.hip`
objects were compiled WITH the loop vectoriser (packed f32 arithmetic: the form the build guard
now refuses everywhere).  Links a second libbtr_pointnet2.so in /tmp, benches `--workload gf`
and `fsb` with each (the in-tree libraries of the box's scratch copy are swapped and restored).  Usage (GPU box): ab_vectorize.py"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from backtoreality_amd import build  # noqa: E402

build.build()
plan = build._object_plan()
alt_dir = "/tmp/btr_vec_lib"
shutil.rmtree(alt_dir, ignore_errors=True)
os.makedirs(alt_dir)
flags = [f for f in build.HIPCC_FLAGS if f != "-fno-vectorize"]
swap = {}
for name in ("attention.hip", "decoder.hip", "gf_loss.hip", "votenet_loss.hip"):
    src = os.path.join(build.CSRC, name)
    obj = os.path.join(alt_dir, name[:-4] + ".o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", src, "-o", obj])
    swap[src] = obj
    print(name, "packed f32 sites with the vectoriser:", len(build.packed_f32_sites(obj)))
bid = build._build_id_object(plan)
for mode, lib in build.LIB_NAMES.items():
    objs = [swap.get(s, o) for s, m, o in plan if m is None or m == mode] + [bid]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=" + build.ARCH, "-shared",
                           "-fPIC"] + objs + ["-o", os.path.join(alt_dir, lib)])


def bench(workload):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload,
                          "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                          "--no-sequential"], capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    return d["ms_per_step"], d["host_enqueue_ms_per_step"]


# (the A/B swaps the in-tree libraries of THIS copy of the repo -- on the GPU box a scratch
# copy -- and puts the shipped ones back)
keep_dir = "/tmp/btr_shipped_lib"
shutil.rmtree(keep_dir, ignore_errors=True)
os.makedirs(keep_dir)
for lib in build.LIB_NAMES.values():
    shutil.copy(os.path.join(build.LIB_DIR, lib), os.path.join(keep_dir, lib))
try:
    for w in ("gf", "fsb"):
        for i in range(2):
            for tag, d in (("no packed f32 (shipped)", keep_dir), ("loop vectoriser on     ", alt_dir)):
                for lib in build.LIB_NAMES.values():
                    shutil.copy(os.path.join(d, lib), os.path.join(build.LIB_DIR, lib))
                ms, host = bench(w)
                print("%-4s %s  %.3f ms  host %.2f" % (w, tag, ms, host))
                sys.stdout.flush()
finally:
    for lib in build.LIB_NAMES.values():
        shutil.copy(os.path.join(keep_dir, lib), os.path.join(build.LIB_DIR, lib))
