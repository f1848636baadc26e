#!/usr/bin/env python3
"""Turns the PMC passes of tools/collect_profiles.sh into profiles/traffic_{orb,loftr}.json (HBM bytes per launch of
each bench stage, the `traffic` field of bench.py's roofline object) and prints a per-kernel table.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half of the bytes of wide coalesced
streaming reads, so the read side is doubled before it is added to WRITE_SIZE; both raw values are kept in the file."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# r04: ONE k_walk launch (threshold units + strips of all eight levels) is the stage "pyramid_fast"; "fast_nms" is the
# check and the dense redo
STAGE = {"k_resize": "pyramid", "k_fast": "fast_nms", "k_walk": "pyramid_fast",
         "k_fast_check": "fast_nms", "k_fast_redo": "fast_nms", "k_thr_harris": "select_harris", "k_select": "select_harris",
         "k_describe": "orient_describe", "k_match": "match", "k_conv": "backbone_convs", "k_tokens": "backbone_convs", "k_out_tokens": "backbone_convs",
         "k_block8": "backbone_convs", "k_block16": "backbone_convs", "k_block8x": "backbone_convs",
         "k_block16x": "backbone_convs", "k_convx": "backbone_convs", "k_down16x": "backbone_convs",
         "k_strip8x": "backbone_convs", "k_stem_strip8x": "backbone_convs", "k_convx2": "backbone_convs", "k_strip16x": "backbone_convs", "k_strip32x": "backbone_convs",
         "k_down32x": "backbone_convs",
         "k_attn_kv": "transformer", "k_attn_update": "transformer", "k_attn_kv_x": "transformer",
         "k_attn_update_x": "transformer", "k_scale_feats": "match_head", "k_harris_flat": "select_harris",
         "k_pair_bound": "match_head", "k_sim_single": "match_head", "k_sim_finish": "match_head",
         "k_sim_cand3": "match_head", "k_conf_cand": "match_head", "k_row_limits": "match_head",
         "k_sim_stats": "match_head", "k_sim_stats3": "match_head", "k_conf_mask": "match_head", "k_decode": "match_head"}


def one_collection(d):
    """The counter CSV of ONE profiled process.  gpurun merges what a call wrote into the local gpurun_out/, it never
    deletes: a pass directory that was collected more than once holds one <pid>_counter_collection.csv per collection,
    and summing them multiplies every figure (the r04 records were 3x one pass).  More than one file is an error here;
    tools/collect_profiles.sh's caller clears the local directory first (tools/pull_clean.sh)."""
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    if not files:
        raise SystemExit("no counter_collection.csv under %s" % d)
    if len(files) > 1:
        raise SystemExit("%d counter collections under %s (%s): stale pulls of earlier runs -- remove the local "
                         "directory, collect once, run again" % (len(files), d, ", ".join(os.path.basename(f) for f in files)))
    return files[0]


def load(d, steps):
    """sum of a counter over all dispatches of a kernel in ONE collection, divided by the number of bench steps (incl.
    warmup)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in [one_collection(d)]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("msf::", "").replace("void ", "")
            if k.startswith("_ZN3msf"):          # a name rocprofv3 left mangled: _ZN3msf<len><identifier>...
                digits = ""
                rest = k[len("_ZN3msf"):]
                while rest and rest[0].isdigit():
                    digits, rest = digits + rest[0], rest[1:]
                if digits:
                    k = rest[:int(digits)]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]) / steps
    return acc


MFMA_DERIVATION = ("sum over the stage's dispatches of SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): "
                   "SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the 1024 SIMDs, the cycles in which a matrix instruction "
                   "executes; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles; both from ONE rocprofv3 --pmc pass "
                   "(<which>_write), so the fraction is matrix-pipe busy SIMD-cycles over available SIMD-cycles")


def mfma_busy(write):
    """per stage and per kernel: MFMA-busy SIMD-cycles over available SIMD-cycles (None without the counters)"""
    num, den, per_kernel = collections.defaultdict(float), collections.defaultdict(float), {}
    for k, c in write.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"] <= 0:
            continue
        avail = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        per_kernel[k] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / avail, 4)
        st = STAGE.get(k)
        if st:
            num[st] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
            den[st] += avail
    if not den:
        return None
    out = {s: round(num[s] / den[s], 4) for s in den}
    out["_per_kernel"] = {k: v for k, v in per_kernel.items() if v > 0}
    out["_derivation"] = MFMA_DERIVATION
    return out


def main(tag="r05", steps=6):
    base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    steps = int(steps)
    for which in ("orb", "orb_vga", "loftr", "loftr_f32"):
        if not os.path.isdir(os.path.join(base, which + "_fetch")):
            continue
        fetch, write = load(os.path.join(base, which + "_fetch"), steps), load(os.path.join(base, which + "_write"), steps)
        stages = collections.defaultdict(lambda: {"fetch_kb_raw": 0.0, "write_kb": 0.0})
        for k in sorted(set(fetch) | set(write)):
            st = STAGE.get(k)
            fk, wk = fetch[k].get("FETCH_SIZE", 0.0), write[k].get("WRITE_SIZE", 0.0)
            print("%-6s %-16s FETCH_SIZE %.1f MB (x2 = %.1f)  WRITE_SIZE %.1f MB per step  -> %s"
                  % (which, k, fk / 1024, 2 * fk / 1024, wk / 1024, st))
            if st:
                stages[st]["fetch_kb_raw"] += fk
                stages[st]["write_kb"] += wk
        out = {s: int((2 * v["fetch_kb_raw"] + v["write_kb"]) * 1024) for s, v in stages.items()}
        out["_raw"] = {s: {"FETCH_SIZE_KB": v["fetch_kb_raw"], "WRITE_SIZE_KB": v["write_kb"]} for s, v in stages.items()}
        mb = mfma_busy(write)
        if mb:
            out["_mfma_busy"] = mb
        # the bench arguments the passes were run with (tools/collect_profiles.sh)
        out["_pairs_per_gpu"] = {"orb": 1024, "orb_vga": 4096}.get(which, 256)
        out["_width"] = 1280 if which == "orb" else 640
        # the kernel sources the passes were measured on: bench.py marks the record stale when they have changed since
        from bench import csrc_digest
        out["_csrc_sha16"] = csrc_digest()
        try:
            import subprocess
            out["_commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            out["_commit"] = None
        out["_note"] = "HBM bytes per bench step (launch of the stage): 2 * FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, " + tag
        json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % which), "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])
