"""Build libnmfx.so (HIP, gfx950 only) in-tree with hipcc.

    python -m nmf_amd.build            # incremental
    python -m nmf_amd.build --force

hipcc cross-compiles for gfx950 without a GPU; the .so lands in nmf_amd/lib/ so
it travels with the source tree (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "lib", "libnmfx.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-ffp-contract=off"]
if os.environ.get("NMFX_EXTRA_DEFS"):          # e.g. "-DNMFX_NNLS_STATS" for tools/anls_perf.py --stats
    FLAGS.extend(os.environ["NMFX_EXTRA_DEFS"].split())


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


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = _hipcc()
    srcs = sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))
    hdrs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "nmfx.h"))
    # a change of flags (NMFX_EXTRA_DEFS experiments) rebuilds everything, in both directions
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    jobs = []
    for s in srcs:
        src = os.path.join(SRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def run(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        p = subprocess.run(cmd, capture_output=True, text=True)
        return job, p

    failed = False
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for (src, obj), p in ex.map(run, jobs):
            if verbose and (p.stderr.strip() or p.returncode):
                sys.stderr.write(p.stderr)
            if p.returncode:
                failed = True
                sys.stderr.write(f"hipcc failed: {src}\n")
    if failed:
        raise RuntimeError("libnmfx build failed")
    with open(stamp, "w") as f:
        f.write(flags_now)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr)
            raise RuntimeError("libnmfx link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
