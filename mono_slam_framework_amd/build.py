"""In-tree native builds: libmsf.so (hipcc, gfx950) and libmsf_synth.so (gcc, host only)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libmsf.so")
SYNTH = os.path.join(_HERE, "libmsf_synth.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SOURCES = ["msf_abi.cpp", "msf_multi.cpp", "msf_gather.cpp", "weights_io.cpp", "orb_kernels.hip", "loftr_kernels.hip", "pack_kernels.hip",
               "ransac_kernels.hip"]
# -ffp-contract=off + correctly rounded f32 divide: the few f32 steps inside ORB
# (Harris response, fastAtan2, pattern rotation) must round exactly like the CPU.
# -mllvm -amdgpu-mfma-vgpr-form: MFMA results straight into VGPRs (the default put the accumulators of k_attn_update_x and
# the similarity kernels into AGPRs and copied them out with 12-52 v_accvgpr_read_b32 per tile; no kernel is near 256 VGPRs).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form",
             "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(ROOT, "include", "msf_abi.h"))
    return d


def ensure_synth(force=False):
    src = os.path.join(CSRC, "synth.c")
    if force or _stale(SYNTH, [src]):
        if not os.path.exists(SYNTH) or os.access(os.path.dirname(SYNTH), os.W_OK):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-o", SYNTH, src])
    return SYNTH


def build_lib(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force or _stale(LIB, _deps()):
        cmd = [HIPCC] + HIP_FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB] + srcs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines=(), flags=()):
    """Developer A/B builds: the same sources with extra -D flags into libmsf_<name>.so next to libmsf.so (the library
    finds the LoFTR weights relative to itself; git-ignored, travels with the gpurun snapshot); selected at run time
    with MSF_LIB_PATH."""
    out = os.path.join(_HERE, "libmsf_%s.so" % name)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [HIPCC] + HIP_FLAGS + list(flags) + ["-D" + d for d in defines] + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out] + srcs + ["-ldl"]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def lib_path():
    """Path of the library; rebuilds when sources are newer and hipcc exists, else fails loudly."""
    override = os.environ.get("MSF_LIB_PATH")      # developer A/B builds (build_variant); never set in production
    if override:
        if not os.path.exists(override):
            raise RuntimeError("MSF_LIB_PATH=%s does not exist" % override)
        return override
    if os.path.exists(LIB) and not _stale(LIB, _deps()):
        return LIB
    if os.path.exists(HIPCC):
        return build_lib()
    if os.path.exists(LIB):
        return LIB
    raise RuntimeError("libmsf.so is missing and hipcc is not available: run __graft_entry__.build()")
