"""In-tree build of the native libraries (no JIT cache: the .so files travel with the repo snapshot).

  libidkpt.so  — HIP kernels + C-ABI (hipcc --offload-arch=gfx950; cross-compiles without a GPU)
  libidkbvh.so — host-side native BVH builder (g++)
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_HERE, "..", "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared", "-fvisibility=hidden"]   # (-munsafe-fp-atomics: native f64 atomic adds for the builder's per-depth sums; no other floating-point atomics exist in the library)
GXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-msse4.1", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-pthread"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False, developer=False):
    """libidkpt.so: the product.  developer=True: libidkpt_dev.so, the same source with -DIDKPT_DEVELOPER — additionally carries the instrumented
    and probe instantiations of the traversal kernel (option "trace_variant"); used by tools/ through IDKPT_LIB_PATH, never by the product path."""
    out = os.path.join(_HERE, "libidkpt_dev.so" if developer else "libidkpt.so")
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + [os.path.join(INCLUDE, f) for f in ("idkpt.h", "idkpt_types.h")]
    if force or _stale(out, srcs):
        cmd = [_hipcc()] + HIPCC_FLAGS + (["-DIDKPT_DEVELOPER"] if developer else []) + ["-o", out, os.path.join(CSRC, "idkpt.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return out


def build_bvh(force=False, verbose=False):
    out = os.path.join(_HERE, "libidkbvh.so")
    srcs = [os.path.join(CSRC, "bvh_builder.cpp")] + [os.path.join(INCLUDE, f) for f in ("idkbvh.h", "idkpt_types.h")]
    if force or _stale(out, srcs):
        cmd = ["g++"] + GXX_FLAGS + ["-o", out, os.path.join(CSRC, "bvh_builder.cpp")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return out


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_bvh(force, verbose)


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
