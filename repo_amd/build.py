"""Build librepo_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension).

    python -m repo_amd.build            # build if sources are newer than the library
    python -m repo_amd.build --force
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "librepo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _deps():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "repo_hip.h")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if _stale(obj, [src] + _deps()):
        subprocess.run([HIPCC, *FLAGS, "-c", src, "-o", obj], check=True)
    return obj


def build(force=False, jobs=None):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in glob.glob(os.path.join(OBJ, "*.o")):
            os.remove(f)
    jobs = jobs or min(6, os.cpu_count() or 1)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB, objs):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
