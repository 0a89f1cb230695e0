"""Build libmpreid_hip.so (the C-ABI library of include/mpreid.h) with hipcc for gfx950.

In-tree build: objects under mp-reid_amd/csrc/_obj/, library at mp-reid_amd/mpreid/libmpreid_hip.so
(git-ignored, travels to the GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
# MPREID_CSRC=<dir>: build another source tree (e.g. the previous round's kernels checked out to a scratch directory, for a
# same-device A/B through MPREID_LIB); needs MPREID_BUILD_TAG so that the product library is never overwritten
CSRC = os.environ.get("MPREID_CSRC") or os.path.join(os.path.dirname(HERE), "csrc")
assert not os.environ.get("MPREID_CSRC") or os.environ.get("MPREID_BUILD_TAG"), "MPREID_CSRC needs MPREID_BUILD_TAG"
# MPREID_BUILD_TAG=<tag>: a second build beside the product one (objects under csrc/_obj_<tag>/) -- e.g. the
# timing-ablation library for a same-device A/B through MPREID_LIB.  Tagged libraries are written to tools/ablation_lib/
# (git-ignored; travels with gpurun), never next to the product library in mpreid/.
_TAG = os.environ.get("MPREID_BUILD_TAG", "")
if os.environ.get("MPREID_ABLATION") and not _TAG:
    _TAG = "abl"          # an ablation build never overwrites the product library
OBJ = os.path.join(os.path.join(os.path.dirname(HERE), "csrc"), "_obj" + ("_" + _TAG if _TAG else ""))
_REPO = os.path.dirname(os.path.dirname(HERE))
LIB = (os.path.join(_REPO, "tools", "ablation_lib", "libmpreid_hip_" + _TAG + ".so") if _TAG
       else os.path.join(HERE, "libmpreid_hip.so"))
SOURCES = ["api.cpp", "distance.hip", "rerank.hip", "gemm_f16.hip", "vit.hip", "evalrank.hip", "preprocess.hip", "conv_f16.hip", "rn50.hip", "rn50_f32.hip"]
# -ffp-contract=off: the rounding sequence of the re-ranking path is part of the contract
# (include/mpreid_numerics.h); fused multiply-adds are written as explicit fmaf().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-x", "hip"]
# MPREID_ABLATION=1: a library that honours the timing-ablation switches (wrong results by design; csrc/common.h)
if os.environ.get("MPREID_ABLATION"):
    FLAGS = FLAGS[:-2] + ["-DMPREID_ABLATION"] + FLAGS[-2:]


# The encoder is a floating-point kernel with a stated tolerance: assuming finite values there removes the
# NaN-canonicalising v_max before every fmaxf.  (distance.hip / rerank.hip keep strict IEEE semantics.)
EXTRA_FLAGS = {"vit.hip": ["-ffinite-math-only", "-fno-signed-zeros"]}


def _deps(src):
    d = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h")]
    inc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
    d += [os.path.join(inc, f) for f in os.listdir(inc)]
    d += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".hpp")]
    return d


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # objects built with other flags (e.g. a timing-ablation build, MPREID_ABLATION=1) must not survive into this build
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(FLAGS) + " | " + repr(sorted(EXTRA_FLAGS.items()))
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        # a different (or unknown) flag set: no object of the old set may be linked.  The objects and the stamp are
        # removed BEFORE anything is compiled, so an interrupted build leaves no stamp and the next one starts over.
        force = True
        for f in os.listdir(OBJ):
            if f.endswith(".o") or f == "flags.txt":
                os.remove(os.path.join(OBJ, f))
    jobs = []
    objs = []
    for src in SOURCES:
        if not os.path.exists(os.path.join(CSRC, src)):
            raise FileNotFoundError(src)
        obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, _deps(src) + [os.path.abspath(__file__)]):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn, file=sys.stderr)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(stamp, "w") as fh:   # only a complete object set of ONE flag set is ever stamped (git-ignored: csrc/_obj/)
        fh.write(flags_now)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
