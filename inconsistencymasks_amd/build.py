"""Compile csrc/*.hip into inconsistencymasks_amd/libimk.so for gfx950 (hipcc cross-compiles without a GPU)."""
import glob
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libimk.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs.  gfx950 has one unified register file; with the default
# heuristic the conv kernels kept their accumulators in AGPRs and paid a v_accvgpr_read/write per accumulator register
# and tile (38 k of them in imk_conv.hip, 13 % of the VALU instructions of the shallow kernel's loop).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-fvisibility=hidden",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def source_id():
    """16 hex digits over the kernel sources, the C-ABI header and the compile flags: identifies the library a committed profile was
    collected with (bench.py prints a replayed value only when this equals the running tree's)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.h")))
    for f in files + [os.path.join(PKG, "..", "include", "imk.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "imk.h")]
    objdir = os.path.join(PKG, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if s.endswith(".cpp"):
                cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]      # zlib: the host PNG codec (imk_png.cpp)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(verbose=True))
