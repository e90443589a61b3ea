"""Build c3dgs_amd/libc3dgs_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m c3dgs_amd.build [--force] [--verbose]

Per-file flags matter: preprocess.hip / backward_preprocess.hip are compiled with -ffp-contract=off
because radii, tile rectangles and depth bits feed bit-exact integer tile keys (see csrc/gsmath.hpp).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libc3dgs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc", "-DNDEBUG"]
SOURCES = {
    "c_abi.hip": [],
    "preprocess.hip": ["-ffp-contract=off"],
    "backward_preprocess.hip": ["-ffp-contract=off"] + os.environ.get("C3DGS_BWDPRE_FLAGS", "").split(),
    "binning.hip": [],
    "radix_sort.hip": os.environ.get("C3DGS_SORT_FLAGS", "").split(),
    # SLP packing into v_pk_*_f32 costs register shuffles in the blend loops and keeps DPP adds from fusing
    "render.hip": os.environ.get("C3DGS_RENDER_FLAGS", "-fno-slp-vectorize").split(),
    # MFMA accumulators in VGPRs (no v_accvgpr_read per value in the top-2 update of the search kernel)
    "vq.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"] + os.environ.get("C3DGS_VQ_FLAGS", "").split(),
    "draws.hip": [],
    "loss.hip": [],
    "encode.hip": [],
    "adam.hip": ["-ffp-contract=off"],
    "qat.hip": ["-ffp-contract=off"],
    "probe.hip": [],            # measurement-only kernels (PMC calibration), see csrc/probe.hip
}
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "gsmath.hpp"),
           os.path.join(HERE, "..", "include", "c3dgs_hip.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(name, extra, verbose):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name.replace(".hip", ".o"))
    cmd = [HIPCC] + COMMON + extra + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


# Test-only variants of the library: same sources, extra flags for some files; objects of untouched files are shared with
# the product build. "spin1": every look-back of the radix sorts gives up after ONE poll, which forces the time-out path
# that tests/test_sort_gpu.py::test_sort_timeout_is_not_silent exercises (loaded through C3DGS_LIB_PATH in a child process).
# "lanes": the blend kernels count how many pixel lanes use each (wave, Gaussian) pair (tools/lane_efficiency.py).
VARIANTS = {"spin1": {"radix_sort.hip": ["-DC3DGS_OS_SPIN_LIMIT=1u"]},
            "lanes": {"render.hip": ["-DC3DGS_COUNT_LANES", "-fno-slp-vectorize"]},
            # "bwdtime": render_backward sums the shader clock per phase (staging / list compaction / group loop / flush)
            "bwdtime": {"render.hip": ["-DC3DGS_BWD_TIMING", "-fno-slp-vectorize"]},
            # (timing-only ablations of render_backward, WRONG gradients, are built by hand: C3DGS_RENDER_FLAGS="-fno-slp-vectorize
            #  -DC3DGS_BWD_ABLATE=1|2|3" python -m c3dgs_amd.build; bit 0 = no partial-sum stores, bit 1 = cache-resident record gathers)
            # "ostime": the digit passes of the onesweep sorts stamp the shader clock at their phase boundaries (tools/sort_phases.py)
            "ostime": {"radix_sort.hip": ["-DC3DGS_OS_TIMING"]}}


def build_variant(name, verbose=False):
    build(verbose=verbose)
    odir = os.path.join(HERE, "build", "variant_" + name)
    os.makedirs(odir, exist_ok=True)
    lib = os.path.join(HERE, f"libc3dgs_hip_{name}.so")
    objs, rebuilt = [], False
    for src, extra in SOURCES.items():
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        if src in VARIANTS[name]:
            obj = os.path.join(odir, src.replace(".hip", ".o"))
            if _stale(obj, [os.path.join(CSRC, src)] + HEADERS + [os.path.abspath(__file__)]):
                cmd = [HIPCC] + COMMON + extra + VARIANTS[name][src] + ["-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                rebuilt = True
        objs.append(obj)
    if rebuilt or _stale(lib, objs):
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    deps_common = HEADERS + [os.path.abspath(__file__)]
    todo, objs = [], []
    for name, extra in SOURCES.items():
        obj = os.path.join(OBJ, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, name)] + deps_common):
            todo.append((name, extra))
    if todo:
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            list(ex.map(lambda t: _compile(t[0], t[1], verbose), todo))
    if force or todo or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
    if "--variants" in sys.argv:
        for v in VARIANTS:
            print(build_variant(v, verbose="--verbose" in sys.argv or "-v" in sys.argv))
