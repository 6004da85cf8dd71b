"""Builds semireward_amd/libsrhip.so (HIP, gfx950 only) in-tree with hipcc.  No torch dependency."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libsrhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-value"]
if os.environ.get("SRHIP_EXTRA_FLAGS"):          # A/B builds on the GPU box (e.g. -DSRHIP_SHFL_REDUCE); never set for the shipped library
    FLAGS += os.environ["SRHIP_EXTRA_FLAGS"].split()
if os.environ.get("SRHIP_TUNING_BUILD"):          # extra diagnostic kernel variants (tools/microbench.py); never shipped
    FLAGS.append("-DSRHIP_TUNING")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        if verbose:
            print("compiled", os.path.basename(s), flush=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
