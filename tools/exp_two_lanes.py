#!/usr/bin/env python3
"""Development experiment (not product, not bench): what would two handles on two streams buy?

A step of either matcher is a chain of dependent launches on ONE stream: every kernel boundary drains the chip (the
last round of workgroups runs on a partly empty machine) before the next kernel starts.  Two handles that each take
half of the pairs, on streams of their own, let one chain's tails fill with the other chain's workgroups.  This
script times  (a) one handle, P pairs per step  against  (b) two handles, P/2 pairs each, steps enqueued alternately
and (c) two handles of P pairs taking alternate steps.

usage: python tools/exp_two_lanes.py [loftr|orb] [--pairs P] [--steps K]
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from mono_slam_framework_amd import _lib, synth  # noqa: E402
from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matcher", nargs="?", default="loftr")
    ap.add_argument("--pairs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    a = ap.parse_args()
    loftr = a.matcher == "loftr"
    W = a.width or (640 if loftr else 1280)
    H = a.height or (480 if loftr else 720)
    P = a.pairs or (256 if loftr else 1024)
    cap = 1024
    dev = torch.device("cuda:0")
    A, B = synth.synth_batch(0, P, W, H, mode=1 if loftr else 0, threads=16)
    dA, dB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)

    def make(n):
        if loftr:
            return DNNFeatureMatcher(threshold=0.15, device=0, max_batch_pairs=n)
        return FeatureMatcher(0.6, W, H, device=0, max_batch_pairs=n)

    def bufs(n):
        return (torch.zeros((n, cap, 4), dtype=torch.int32, device=dev), torch.zeros((n,), dtype=torch.int32, device=dev))

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    # (a) one handle
    fm = make(P)
    o, c = bufs(P)
    s = torch.cuda.Stream(device=dev)

    def one():
        for _ in range(a.steps):
            fm.match_batch_device(dA, dB, o, c, stream=s.cuda_stream)

    one()
    ta = timed(one)
    ref_cnt = c.clone()
    print(f"(a) one handle, {P} pairs/step: {ta / a.steps * 1e3:.3f} ms/step, {P * a.steps / ta:.0f} pairs/s", flush=True)
    # (c) two handles of P pairs, alternate steps
    fm2 = make(P)
    o2, c2 = bufs(P)
    s2 = torch.cuda.Stream(device=dev)

    def alt():
        for k in range(a.steps):
            if k & 1:
                fm2.match_batch_device(dA, dB, o2, c2, stream=s2.cuda_stream)
            else:
                fm.match_batch_device(dA, dB, o, c, stream=s.cuda_stream)

    alt()
    tc = timed(alt)
    assert torch.equal(c2, ref_cnt) and torch.equal(c, ref_cnt)
    print(f"(c) two handles x {P} pairs, alternate steps: {tc / a.steps * 1e3:.3f} ms/step, {P * a.steps / tc:.0f} pairs/s", flush=True)
    del fm2, o2, c2
    # (b) two handles of P/2 pairs, every step split
    h = P // 2
    fa, fb = make(h), make(h)
    oa, ca = bufs(h)
    ob, cb = bufs(h)

    def split():
        for _ in range(a.steps):
            fa.match_batch_device(dA[:h], dB[:h], oa, ca, stream=s.cuda_stream)
            fb.match_batch_device(dA[h:], dB[h:], ob, cb, stream=s2.cuda_stream)

    split()
    tb = timed(split)
    assert torch.equal(torch.cat([ca, cb]), ref_cnt)
    print(f"(b) two handles x {h} pairs, split steps: {tb / a.steps * 1e3:.3f} ms/step, {P * a.steps / tb:.0f} pairs/s", flush=True)


if __name__ == "__main__":
    main()
