"""Build recipe for the native libraries (hipcc, gfx950 only).  Used by __graft_entry__.build() and `python -m hydrochrono_amd.build`.

  hydrochrono_amd/lib/libhydrochrono_amd.so   C ABI + HIP kernels (include/hydrochrono_amd.h): the RELEASE library
  hydrochrono_amd/lib/hc_kernels.co           the kernels as a stand-alone gfx950 code object (direct AQL dispatch of the step path)
  hydrochrono_amd/lib/libhc_bemio.so          optional BEMIO-HDF5 reader / result-file writer (only where libhdf5 is installed); depends on
                                              neither flavour of the main library
  hydrochrono_amd/lib/libhydrochrono_amd_tuning.so + hc_kernels_tuning.co
                                              the same sources with -DHC_TUNING: the sweep / A-B / fault-injection switches (HC_TUNE_INT
                                              in csrc/hc_internal.hpp) and the kernel variants that were measured and not taken
                                              (EXPERIMENTS.md).  The tests that need such a switch load this one (capi.use_flavor).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
MAIN_LIB = os.path.join(LIBDIR, "libhydrochrono_amd.so")
TUNING_LIB = os.path.join(LIBDIR, "libhydrochrono_amd_tuning.so")
BEMIO_LIB = os.path.join(LIBDIR, "libhc_bemio.so")

SOURCES = ["hc_kernels.hip", "hc_runtime.cpp", "hc_step.cpp", "hc_pass.cpp", "hc_setup.cpp", "hc_query.cpp", "hc_direct.cpp", "hc_host_math.cpp", "hc_yaml.cpp",
           "hc_eta_fft.cpp"]
# kernel-argument preload: the leading scalar / pointer arguments of a kernel arrive in scalar registers with the wave (up to 16 words:
# added_mass_mv_tagged_kernel starts without a single argument load; the finalize_pre_kernel experiment of the tuning build, EXPERIMENTS.md
# round 6); kernels whose first argument is a struct -- the step path's -- are unaffected.  The code object only: the library's embedded
# copies go through HIP launches.
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
KERNEL_CO = os.path.join(LIBDIR, "hc_kernels.co")  # the same kernels as a stand-alone code object, for the direct AQL dispatch (hc_direct.hpp)
TUNING_CO = os.path.join(LIBDIR, "hc_kernels_tuning.co")
HEADERS = ["hc_kernels.hpp", "hc_context.hpp", "hc_internal.hpp", "hc_host_math.hpp", "hc_limits.hpp", "hc_plan.hpp", "hc_history.hpp", "hc_direct.hpp", "hc_fanout.hpp", "hc_h5data.hpp", os.path.join(ROOT, "include", "hydrochrono_amd.h"),
           os.path.join(ROOT, "include", "hydrochrono_amd_host.h"), os.path.join(ROOT, "include", "hydrochrono_amd_yaml.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _find_hdf5():
    for prefix in (os.environ.get("HDF5_ROOT"), "/opt/conda", "/usr", "/usr/local"):
        if not prefix:
            continue
        inc = os.path.join(prefix, "include", "hdf5.h")
        for libdir in ("lib", "lib64", "lib/x86_64-linux-gnu"):
            lib = os.path.join(prefix, libdir, "libhdf5.so")
            if os.path.exists(inc) and os.path.exists(lib):
                return os.path.join(prefix, "include"), os.path.join(prefix, libdir)
    return None


def _build_flavor(lib, kernel_co, objdir, defines, force, verbose):
    """One flavour of the main library: one object per source, compiled side by side (the kernels take most of the time), one link,
    and the kernels once more as a stand-alone code object.  Returns the processes still running (the code object's compile)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    headers = [d for d in deps if d.endswith((".hpp", ".h"))]
    kernel_src = os.path.join(CSRC, "hc_kernels.hip")
    pending = []
    if force or _newer(kernel_co, [kernel_src] + headers):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "--no-gpu-bundle-output", "-Wno-unused-result"] + PRELOAD + defines + [
            "-I", os.path.join(ROOT, "include"), kernel_src, "-o", kernel_co]
        if verbose:
            print(" ".join(cmd))
        pending.append((cmd, subprocess.Popen(cmd)))
    if force or _newer(lib, deps):
        os.makedirs(objdir, exist_ok=True)
        base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-result"] + defines + ["-I", os.path.join(ROOT, "include")]
        objs = [os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o") for src in srcs]
        todo = [(src, obj) for src, obj in zip(srcs, objs) if force or _newer(obj, [src] + headers)]
        procs = []
        for src, obj in todo:
            cmd = base + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        # -Bsymbolic: the library's calls of its own entry points stay inside it whichever flavour was loaded first
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"] + objs + ["-o", lib, "-ldl", "-lrocfft", "-lhsa-runtime64"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return pending


def build(force=False, verbose=False, tuning=True):
    os.makedirs(LIBDIR, exist_ok=True)
    pending = _build_flavor(MAIN_LIB, KERNEL_CO, os.path.join(LIBDIR, "obj"), [], force, verbose)
    if tuning:
        pending += _build_flavor(TUNING_LIB, TUNING_CO, os.path.join(LIBDIR, "obj_tuning"), ["-DHC_TUNING"], force, verbose)
    for cmd, pr in pending:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    h5 = _find_hdf5()
    bemio_src = os.path.join(CSRC, "hc_bemio.cpp")
    if h5 and (force or _newer(BEMIO_LIB, [bemio_src, os.path.join(CSRC, "hc_h5data.hpp")])):
        inc, lib = h5
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", bemio_src, "-o", BEMIO_LIB, "-I", inc, "-L", lib, "-lhdf5", f"-Wl,-rpath,{lib}"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd)
        if r.returncode != 0:
            print("warning: BEMIO-HDF5 reader not built (hc_load_bemio_h5 will return HC_ERR_UNSUPPORTED)", file=sys.stderr)
    return MAIN_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tuning="--no-tuning" not in sys.argv))
