"""Build libdcvic_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python dc_vic_amd/csrc/build.py [--force]

Objects are cached under dc_vic_amd/csrc/_obj and rebuilt when a source or header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "libdcvic_hip.so")
SOURCES = ["conv.hip", "conv3x3.hip", "wino.hip", "wino44.hip", "thin.hip", "conv_async.hip", "conv_async16.hip", "conv1x1.hip", "gemm.hip", "attn.hip", "norm.hip", "ew.hip", "swin.hip", "vq.hip", "rate.hip", "train.hip", "error.cpp", "host_entropy.cpp"]
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(HERE, "conv_common.h"), os.path.join(ROOT, "include", "dcvic.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result",
         f"-I{os.path.join(ROOT, 'include')}", f"-I{HERE}"]


# per-file extras.  wino.hip: hipcc's SLP vectoriser packs the input transform's adds into v_pk_add_f32, which costs MFMA issue
# time beside the matrix pipe (MI355X_MICROARCH "packed f32 VALU ... an anti-lever beside MFMAs")
# thin.hip: the packed form of its fmaf chains needs a v_mov per misaligned register pair (491 of them in thin_cout_kernel<3>)
EXTRA_FLAGS = {"wino.hip": ["-fno-slp-vectorize"], "wino44.hip": ["-fno-slp-vectorize"], "thin.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_m = max(os.path.getmtime(h) for h in HEADERS)
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            lang = ["-x", "hip"] if s.endswith(".hip") else []
            jobs.append([_hipcc(), *FLAGS, *EXTRA_FLAGS.get(s, []), *lang, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[dcvic build]", os.path.basename(cmd[-3]), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-lpthread"])
        if verbose:
            print("[dcvic build] linked", OUT, flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
