#!/usr/bin/env python3
"""Static audit of the compiled kernels (runs in the build container, no GPU): for every kernel of csrc/*.hip the register
count, LDS, scratch, the waves per SIMD its registers allow, and the number of `s_waitcnt vmcnt(0)` inside loops.

Two stalls of round 3 were visible here and in no counter: a vector load of a constant table issued behind a prefetch and
waited for with vmcnt(0) (the counter is in-order: the wait covered the prefetch; k_describe), and a five-wave workgroup
shape that left 46 % of the wave slots empty at four waves per SIMD (k_sim_stats3).  A vmcnt(0) inside the main loop of a
kernel that prefetches is the first thing to look at; so is a workgroup whose wave count is not a multiple of 4.

usage: tools/asm_audit.py [orb|loftr|pack|ransac ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mono_slam_framework_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-S"]


def hipcc():
    """the compiler build.py uses: $HIPCC, else hipcc on PATH, else /opt/rocm/bin/hipcc (None if there is none)"""
    cand = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    return cand if (os.path.isabs(cand) and os.path.exists(cand)) or shutil.which(cand) else None


def audit(src):
    cc = hipcc()
    if cc is None:
        raise FileNotFoundError("no hipcc ($HIPCC, PATH, /opt/rocm/bin)")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([cc] + FLAGS + ["-o", out, src], stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed on %s:\n%s" % (cc, src, r.stderr[-4000:]))
        lines = open(out).read().split("\n")
    stats, cur, inloop = {}, None, False
    for ln in lines:
        m = re.match(r"^(_ZN3msf\w+):", ln)
        if m:
            cur = m.group(1)
            stats[cur] = {"vm0": 0, "vm0_loop": 0, "loads": 0, "lines": 0, "scratch_ops": 0}
            inloop = False
            continue
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", ln)
        if m:
            cur = m.group(1)
            stats.setdefault(cur, {"vm0": 0, "vm0_loop": 0, "loads": 0, "lines": 0, "scratch_ops": 0})
            continue
        if cur is None:
            continue
        st = stats[cur]
        for key, pat in (("vgpr", r"\.amdhsa_next_free_vgpr (\d+)"), ("lds", r"\.amdhsa_group_segment_fixed_size (\d+)"),
                         ("scratch", r"\.amdhsa_private_segment_fixed_size (\d+)")):
            m = re.search(pat, ln)
            if m:
                st[key] = int(m.group(1))
        if ".end_amdhsa_kernel" in ln or "s_endpgm" in ln:
            if "s_endpgm" in ln:
                cur = None
            continue
        st["lines"] += 1
        if "Loop Header" in ln or "in Loop:" in ln:
            inloop = True
        if re.match(r"^\.LBB\d+_\d+:\s*$", ln):
            inloop = False
        if "s_waitcnt" in ln and "vmcnt(0)" in ln:
            st["vm0"] += 1
            st["vm0_loop"] += 1 if inloop else 0
        if "global_load" in ln or "buffer_load" in ln:
            st["loads"] += 1
        # scratch-memory instructions actually emitted (the descriptor's private-segment size also counts the frame the
        # compiler reserves behind scalar-register spills to vector lanes, which no instruction ever touches)
        if re.match(r"^\s*scratch_(load|store)", ln):
            st["scratch_ops"] += 1
    return stats


def demangle(k):
    m = re.match(r"_ZN3msf(\d+)", k)
    if not m:
        return k
    n = int(m.group(1))
    rest = k[len(m.group(0)):]
    return rest[:n] + ("<" + rest[n:n + 24] + ">" if rest[n:n + 1] == "I" else "")


if __name__ == "__main__":
    want = sys.argv[1:] or ["orb", "loftr", "pack", "ransac"]
    for w in want:
        src = os.path.join(CSRC, w + "_kernels.hip")
        print("== %s" % os.path.basename(src))
        print("  %-46s %5s %7s %7s %10s %6s %14s" % ("kernel", "VGPRs", "LDS B", "scratch", "waves/SIMD", "loads", "vmcnt(0) [loop]"))
        for k, st in audit(src).items():
            if "vgpr" not in st:
                continue
            v = st["vgpr"]
            alloc = (v + 7) // 8 * 8
            print("  %-46s %5d %7d %7d %10d %6d %8d [%d]" % (demangle(k)[:46], v, st.get("lds", 0), st.get("scratch", 0),
                                                            min(8, 512 // max(alloc, 1)), st["loads"], st["vm0"], st["vm0_loop"]))
