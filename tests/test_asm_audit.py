"""Static checks on the compiled kernels (build container, no GPU): the register / scratch / wait properties the
round-3 occupancy and latency fixes rest on.  tools/asm_audit.py compiles csrc/*_kernels.hip to gfx950 assembly and
reads the kernel descriptors; a compiler or code change that silently halves a kernel's occupancy, puts a kernel's
struct copy into scratch memory or re-introduces a full vector-memory wait into a prefetching loop fails here."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import asm_audit  # noqa: E402

pytestmark = pytest.mark.skipif(asm_audit.hipcc() is None, reason="needs hipcc (cross-compiles without a GPU)")


def _audit(which):
    import asm_audit
    return asm_audit.audit(os.path.join(asm_audit.CSRC, which + "_kernels.hip"))


def _by_prefix(raw, prefix):
    import asm_audit
    return [v for k, v in raw.items() if "vgpr" in v and asm_audit.demangle(k).startswith(prefix)]


def test_orb_kernels_keep_their_occupancy_and_waits():
    raw = _audit("orb")
    walk = _by_prefix(raw, "k_walk")
    assert len(walk) == 1                          # one kernel: threshold units, resizing strips and FAST-only strips
    for st in walk:
        # no scratch-memory instruction: the walker must not spill vector registers or copy its arguments to scratch.
        # (r05: the descriptor reserves 68 B behind the scalar-register spills -- the kernel sits at 100 SGPRs, spilled to
        # vector lanes with v_writelane -- which nothing reads or writes; a real spill shows up as instructions)
        assert st["scratch_ops"] == 0 and st["scratch"] <= 128, "the walker must not spill or copy its arguments to scratch memory"
        assert st["vgpr"] <= 104 and st["lds"] <= 10240, "four waves per SIMD / sixteen waves per CU"
    desc = _by_prefix(raw, "k_describe")
    assert len(desc) == 3
    for st in desc:
        assert st["scratch"] == 0 and st["vgpr"] <= 96
        # the loop's only full vector-memory wait is at its top (where the prefetched patch is consumed) + the epilogue
        assert st["vm0_loop"] <= 2, "a vmcnt(0) inside k_describe's key-point loop waits for the next patch's prefetch"


def test_loftr_streaming_kernels_fit_two_workgroups_per_cu():
    raw = _audit("loftr")
    for name, max_vgpr, max_scratch in (("k_down16x", 128, 0), ("k_strip32x", 128, 64), ("k_strip16x", 128, 0),
                                        ("k_stem_strip8x", 80, 32), ("k_sim_single", 128, 0), ("k_sim_cand3", 128, 0)):
        sts = _by_prefix(raw, name)
        assert sts, name
        for st in sts:
            # (128 registers = two 8-wave workgroups per CU; k_stem_strip8x runs three: 80 registers, 20 B of spills)
            assert st["vgpr"] <= max_vgpr, "%s: %d registers (fewer workgroups per CU than it is built for)" % (name, st["vgpr"])
            assert st["scratch"] <= max_scratch, "%s spills %d B" % (name, st["scratch"])
