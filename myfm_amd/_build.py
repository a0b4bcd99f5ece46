"""In-tree build of the native parts: libmyfm_hip.so (hipcc, gfx950) and the pybind11 module _myfm
(g++, linked against it). Built artefacts stay next to the package so they travel with the tree."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
HIP_LIB = os.path.join(HERE, "libmyfm_hip.so")
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
PYMOD = os.path.join(HERE, "_myfm" + EXT_SUFFIX)

# translation units of libmyfm_hip.so and the headers each of them includes (a unit is recompiled when one of them is newer than
# its object file; the objects live in csrc/_obj/, git-ignored)
HIP_UNITS = {
    "mfm_hip.hip": ["mfm_common.hpp", "mfm_wave.hpp", "mfm_policies.hpp", "mfm_kernels.hpp", "mfm_mf_kernels.hpp", "mfm_res.hpp", "mfm_res_plan.hpp",
                    "mfm_plan.hpp", "mfm_block_kernels.hpp", "mfm_tasks.hpp", "mfm_predict.hpp", "mfm_rng.hpp", "mfm_mtjump.hpp", "mfm_cell.hpp",
                    "mfm_chain_api.hpp", "mfm_rng_state.hpp", "mfm_latent_api.hpp", "mfm_latent_host.hpp"],
    "mfm_latent.hip": ["mfm_common.hpp", "mfm_rng_state.hpp", "mfm_latent_api.hpp"],
    "mfm_cell.hip": ["mfm_common.hpp", "mfm_wave.hpp", "mfm_cell.hpp"],
    "mfm_chain.hip": ["mfm_common.hpp", "mfm_wave.hpp", "mfm_policies.hpp", "mfm_chain_api.hpp", "mfm_chain_plan.hpp", "mfm_chain_stream.hpp"],
}
OBJ_DIR = os.path.join(CSRC, "_obj")
PYMOD_HEADERS = ["mfm_hostnormals.hpp", "mfm_mtjump.hpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def build_hip(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + INCLUDE, "-I" + CSRC] + \
        os.environ.get("MYFM_AMD_HIPCC_FLAGS", "").split()
    jobs, objs = [], []
    for src, hdrs in HIP_UNITS.items():
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in hdrs] + [os.path.join(INCLUDE, "myfm_hip.h")]
        if force or _newer(obj, deps):
            cmd = [_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = [(cmd, pr.returncode) for cmd, pr in jobs if pr.wait() != 0]  # (the units compile side by side; every job is waited for)
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], " ".join(failed[0][0]))
    if jobs or not os.path.exists(HIP_LIB) or _newer(HIP_LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", HIP_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_pymod(force=False, verbose=False):
    import pybind11

    src = os.path.join(CSRC, "_myfm.cpp")
    deps = [src, os.path.join(INCLUDE, "myfm_hip.h")] + [os.path.join(CSRC, h) for h in PYMOD_HEADERS]  # (linked dynamically)
    if not (force or _newer(PYMOD, deps)):
        return PYMOD
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-I" + pybind11.get_include(),
           "-I" + sysconfig.get_paths()["include"], "-I" + INCLUDE, src, "-o", PYMOD, "-L" + HERE, "-lmyfm_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return PYMOD


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_pymod(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)


def preload_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's). Two HIP
    runtimes in one process do not coexist (the second one sees no GPU), and the row-sharded mode hands this
    library's device buffers to torch.distributed. So when torch is installed, its runtime is loaded first
    and libmyfm_hip.so binds to it; without torch the system ROCm runtime is used. Set
    MYFM_AMD_SYSTEM_HIP=1 to force the system runtime."""
    import ctypes
    import importlib.util

    if os.environ.get("MYFM_AMD_SYSTEM_HIP"):
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(lib):
        return None
    try:
        return ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None
