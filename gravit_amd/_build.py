"""Build libgvt_hip.so (the C-ABI adapter library) and libgvt_hip_exp.so (the same sources with -DGVT_EXPERIMENTS: every
variant that was measured and lost behind its knob, for the knob sweeps and probes) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the tree to
the GPU box.  -ffp-contract=off is part of the contract: the parity-critical arithmetic must not
be fused (see csrc/gvt_device.h).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libgvt_hip.so")
LIB_EXP = os.path.join(HERE, "libgvt_hip_exp.so")
SOURCES = ["api.hip", "lbvh.hip", "trace.hip", "sched.hip", "domain.hip"]


def _headers(experiments):
    """Every header / include a translation unit may pull in: csrc/*.h, csrc/*.inc, include/*.h (+ csrc/experiments/* for the
    experiments build) -- the same set source_hash() hashes, so a stale object can never carry a fresh hash."""
    inc = os.path.join(os.path.dirname(HERE), "include")
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]
    out += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    if experiments:
        exp = os.path.join(CSRC, "experiments")
        out += [os.path.join(exp, f) for f in sorted(os.listdir(exp)) if f.endswith((".h", ".inc"))]
    return out

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]


def source_hash():
    """Hash of the sources a build is made of (csrc/**, include/*.h, bench.py): ties a bench line or a profile to a tree where no .git travels."""
    import hashlib

    root = os.path.dirname(HERE)
    files = [os.path.join(root, "bench.py")]
    for base in (CSRC, os.path.join(root, "include")):
        for d, _, fs in os.walk(base):
            if "build" in os.path.relpath(d, root).split(os.sep):  # csrc/build (objects): judged on the path INSIDE the tree, wherever the tree lies
                continue
            files += [os.path.join(d, f) for f in fs if f.endswith((".hip", ".h", ".inc"))]
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found: cannot build the gfx950 adapter library")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False):
    """The shipped library, or with experiments=True the experiments build (objects in csrc/build/exp)."""
    obj_dir = os.path.join(OBJ, "exp") if experiments else OBJ
    lib_path = LIB_EXP if experiments else LIB
    os.makedirs(obj_dir, exist_ok=True)
    extra = os.environ.get("GVT_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DTRAV_STACK=16); forces a rebuild
    if extra:
        force = True
    if experiments:
        extra = extra + ["-DGVT_EXPERIMENTS"]
    cc = hipcc()
    hdrs = _headers(experiments)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [cc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if force or _stale(lib_path, objs):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib_path] + objs
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if out.returncode != 0:
            raise RuntimeError("link failed:\n%s" % out.stdout.decode(errors="replace"))
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build(force="--force" in sys.argv, verbose=True, experiments=True))
