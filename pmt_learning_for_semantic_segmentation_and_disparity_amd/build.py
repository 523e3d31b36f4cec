"""In-tree build of libsdhip.so (hipcc, gfx950 only).

`python -m pmt_learning_for_semantic_segmentation_and_disparity_amd.build` or
`__graft_entry__.build()` compile every csrc/*.hip to an object file and link
one shared library next to this file.  hipcc cross-compiles without a GPU.
The library links against libamdhip64.so.7 by soname only, so that when it is
dlopen()ed after `import torch` it binds to the HIP runtime PyTorch-ROCm has
already loaded (one runtime per process: streams and device pointers are shared).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libsdhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-mcode-object-version=5", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(HERE, "..", "include")]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "sdhip.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for (src, obj), r in ex.map(cc, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(r.stderr)
                if r.returncode:
                    raise RuntimeError("hipcc failed on %s" % src)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link of libsdhip.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
