#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec (extract + match) of the MI355X-native matcher stage.

A "step" is one pass of the hot path (ORB extract x2 + Hamming 2-NN + ratio test + match-list pack, and for
N > 1 the RCCL gather of the packed match lists to rank 0) over one batch of synthetic 1280x720 pairs that
is resident in HBM before the timed region starts (BASELINE.json configs[1]).  One process per GPU; launched
by torch.distributed.run for N > 1.  Pairs are independent, so ranks shard them with no data-path collective
other than the gather of results ("weak" scaling: every rank owns --pairs pairs).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, exact f32 (MSF_FLAG_LOFTR_F32)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (the default LoFTR path issues v_mfma_f32_16x16x32_bf16)
LOFTR_FLOPS_PER_PAIR = 2.601e9  # conv 2.273 G + matmul 0.328 G (2 x MAC)
LOFTR_CONV_FLOPS_PER_PAIR = 2.273e9


def algorithmic_bytes_per_pair(w, h):
    # 2*W*H u8 reads + <= 2*500*(8+32) B features + <= 500*16 B matches
    return 2 * w * h + 2 * 500 * 40 + 500 * 16


def host_cpu():
    """nproc + CPU model of the box the CPU leg runs on (SURVEY.md 8d asks for both next to every CPU figure)."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a cgroup CPU quota (cpu.max "<quota> <period>") bounds what any number of threads can get
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                if t[0] != "max":
                    quota = float(t[0]) / float(t[1])
            elif int(t[0]) > 0:
                quota = int(t[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        usable = max(1, min(usable, int(quota + 0.999)))
    return {"nproc": os.cpu_count() or 1, "usable": usable, "cgroup_cpu_quota": quota, "model": model}


def _orb_cpu_leg(A, B, ratio, threads, seconds, one, floor_per_thread=1):
    """`threads` Python threads (ctypes drops the GIL), each with its own oracle object, one pair at a time per thread,
    on a sample sized for about `seconds` of wall time from the one-pair calibration `one`.  -> (pairs, wall, results)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orb as oracle_orb
    n = int(min(len(A), max(threads * floor_per_thread, threads * (seconds / one))))

    def work(t):
        orc = oracle_orb.FeatureMatcherOracle(ratio)
        return [(i, orc.MatchFrames(A[i], B[i])) for i in range(t, n, threads)]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        results = [r for part in ex.map(work, range(threads)) for r in part]
    return n, time.perf_counter() - t0, results


def cpu_baseline(args, A, B, gpu_lists):
    """Times the CPU oracle (kind "port") on bounded samples of the same pairs, one pair per thread: the --cpu-threads
    figure of the earlier rounds (`value`), and the two SURVEY.md 8d asks for -- `single_thread` and `all_cores` (every
    hardware thread this process may run on, affinity stated).  Each sample is sized from a one-pair calibration."""
    import numpy as np
    from oracle import orb as oracle_orb
    host = host_cpu()
    cores = max(1, min(host["usable"], args.cpu_threads))
    oracle_orb.lib()
    t0 = time.perf_counter()
    oracle_orb.FeatureMatcherOracle(args.ratio).MatchFrames(A[0], B[0])
    one = max(time.perf_counter() - t0, 1e-4)
    n, dt, results = _orb_cpu_leg(A, B, args.ratio, cores, args.cpu_seconds, one, args.cpu_pairs_per_thread)
    mismatches = 0
    for i, m in results:
        if gpu_lists is not None and not (len(m) == len(gpu_lists[i]) and np.array_equal(m, gpu_lists[i])):
            mismatches += 1
    out = {"value": round(n / dt, 3), "unit": "frame-pairs/sec", "cores": cores, "kind": "port",
           "sample": "%d of the same %dx%d pairs, oracle/orb_oracle.c (scalar C restatement), one pair per thread, "
                     "%d unpinned threads, %.1f s" % (n, args.width, args.height, cores, dt),
           "host": host, "parity_mismatches_vs_gpu": mismatches}
    if args.cpu_extra_seconds > 0:
        n1, dt1, _ = _orb_cpu_leg(A, B, args.ratio, 1, args.cpu_extra_seconds, one)
        out["single_thread"] = {"value": round(n1 / dt1, 3), "cores": 1, "sample": "%d pairs, %.1f s, unpinned" % (n1, dt1)}
        na, dta, _ = _orb_cpu_leg(A, B, args.ratio, host["usable"], args.cpu_extra_seconds, one)
        out["all_cores"] = {"value": round(na / dta, 3), "cores": host["usable"],
                            "sample": "%d pairs, %.1f s, one pair per thread" % (na, dta),
                            "affinity": "%d of %d hardware threads usable by this process (sched_getaffinity, capped by "
                                        "the cgroup CPU quota %s), threads unpinned within that set"
                                        % (host["usable"], host["nproc"], host["cgroup_cpu_quota"])}
    return out


def _loftr_cpu_pairs_parallel(A, B, threshold, threads, seconds, one):
    """one pair per thread: every worker thread sets ITS OpenMP team to 1 (omp_set_num_threads acts on the calling
    thread) and runs whole pairs on an oracle object of its own.  -> (pairs, wall)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import loftr as oracle_loftr
    n = int(min(len(A), max(threads, threads * (seconds / one))))

    def work(t):
        oracle_loftr.set_threads(1)
        orc = oracle_loftr.DNNFeatureMatcherOracle(threshold)
        for i in range(t, n, threads):
            orc.MatchFrames(A[i], B[i])
        return 0

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(work, range(threads)))
    return n, time.perf_counter() - t0


def cpu_baseline_loftr(args, A, B, gpu_lists):
    """LoFTR CPU baseline: the C restatement (OpenMP over its convolutions), pairs one after another (`value`), plus
    `single_thread` (one OpenMP thread) and `all_cores` (one pair per hardware thread, no OpenMP team)."""
    import numpy as np
    from oracle import loftr as oracle_loftr
    host = host_cpu()
    cores = oracle_loftr.set_threads(max(1, min(host["usable"], args.cpu_threads)))
    orc = oracle_loftr.DNNFeatureMatcherOracle(args.threshold)
    t0 = time.perf_counter()
    orc.MatchFrames(A[0], B[0])
    one = max(time.perf_counter() - t0, 1e-4)
    n = int(min(len(A), max(args.cpu_loftr_pairs, args.cpu_seconds / one)))
    t0 = time.perf_counter()
    confs = [orc.run(A[i], B[i])["conf"] for i in range(n)]
    res = [orc.decode(c) for c in confs]
    dt = time.perf_counter() - t0
    extra = {}
    if args.cpu_extra_seconds > 0:
        oracle_loftr.set_threads(1)
        t0 = time.perf_counter()
        orc.MatchFrames(A[0], B[0])
        one1 = max(time.perf_counter() - t0, 1e-4)
        n1 = int(min(len(A), max(1, args.cpu_extra_seconds / one1)))
        t0 = time.perf_counter()
        for i in range(n1):
            orc.MatchFrames(A[i], B[i])
        dt1 = time.perf_counter() - t0
        extra["single_thread"] = {"value": round(n1 / dt1, 3), "cores": 1, "sample": "%d pairs, %.1f s, unpinned" % (n1, dt1)}
        na, dta = _loftr_cpu_pairs_parallel(A, B, args.threshold, host["usable"], args.cpu_extra_seconds, one1)
        extra["all_cores"] = {"value": round(na / dta, 3), "cores": host["usable"],
                              "sample": "%d pairs, %.1f s, one pair per thread (OpenMP team of 1 each)" % (na, dta),
                              "affinity": "%d of %d hardware threads usable by this process (sched_getaffinity, capped by "
                                          "the cgroup CPU quota %s), threads unpinned within that set"
                                          % (host["usable"], host["nproc"], host["cgroup_cpu_quota"])}
        oracle_loftr.set_threads(cores)
    # Parity rule for the lists: identical wherever |conf - threshold| > 1e-3.  An entry that only one side
    # lists is therefore within tolerance iff the restatement's confidence of that (token, token) pair lies within 1e-3
    # of the threshold; anything else is a real mismatch.
    thr, tol = float(args.threshold), 1e-3
    pairs_diff = entries_diff = beyond = 0
    worst = 0.0
    for i, m in enumerate(res):
        a = set(map(tuple, np.asarray(m).reshape(-1, 4).tolist()))
        b = set(map(tuple, np.asarray(gpu_lists[i]).reshape(-1, 4).tolist()))
        d = a ^ b
        if d or len(m) != len(gpu_lists[i]):
            pairs_diff += 1
        for (x1, y1, x2, y2) in d:
            # token t <-> pixel ((t % 40) * 16, (t // 40) * 16) (dnnfeaturematcher.cpp:88-99)
            margin = abs(float(confs[i][(y1 // 16) * 40 + x1 // 16, (y2 // 16) * 40 + x2 // 16]) - thr)
            entries_diff += 1
            worst = max(worst, margin)
            beyond += int(margin > tol)
    out = {"value": round(n / dt, 3), "unit": "frame-pairs/sec", "cores": cores, "kind": "port",
           "sample": "%d of the same 640x480 pairs, oracle/loftr_oracle.c (f32 C restatement, OpenMP with %d unpinned "
                     "threads), %.1f s" % (n, cores, dt),
           "host": host,
           "match_lists_vs_gpu": {"pairs": n, "pairs_with_a_difference": pairs_diff, "entries_on_one_side_only": entries_diff,
                                  "entries_beyond_tolerance": beyond,
                                  "largest_margin_of_such_an_entry": round(worst, 6), "tolerance": tol,
                                  "rule": "lists identical wherever |conf - threshold| > 1e-3 (conf of the CPU restatement)"}}
    out.update(extra)
    return out


ORB_SWITCHES = ("blur_tie_even", "level_size_mul_inv", "blur_kernel_sum256")   # the oracle's / product's open choices


def opencv_probe(args, A, B, gpu_lists):
    """SURVEY.md H1 / BASELINE.md 3.2: if a real OpenCV happens to be importable on this box, run the literal reference
    sequence (cv::ORB::create() defaults, detectAndCompute x2 with an all-255 mask, BFMatcher(NORM_HAMMING).knnMatch(k=2),
    ratio test, int truncation: src/featurematcher.cpp:3-45) on the same pairs and record (a) how the GPU lists compare
    with it and (b) WHICH combination of the restatement's documented open choices (blur rounding, blur
    kernel, level-size formula) reproduces the library -- key-point set, descriptor bytes and match set are scored
    for all 8 combinations, not just the default.  Never required: absent -> {"opencv": "absent"}."""
    try:
        import cv2
    except Exception:
        return {"opencv": "absent"}
    import itertools
    import numpy as np
    from oracle import orb as oracle_orb
    n = min(len(A), 8)
    orb = cv2.ORB_create()
    bf = cv2.BFMatcher(cv2.NORM_HAMMING)
    t0 = time.perf_counter()
    ref = []          # per pair: (features of A, features of B, match list); features = {key: descriptor bytes}
    for i in range(n):
        feats = []
        kd = []
        for im in (A[i], B[i]):
            k, d = orb.detectAndCompute(im, np.full(im.shape, 255, np.uint8))
            d = np.zeros((0, 32), np.uint8) if d is None else np.asarray(d)
            kd.append((k, d))
            feats.append({(int(q.octave) & 255, np.float32(q.pt[0]).tobytes(), np.float32(q.pt[1]).tobytes()): d[j].tobytes()
                          for j, q in enumerate(k)})
        (k1, d1), (k2, d2) = kd
        m = []
        if len(k1) and len(k2) >= 2:
            for pr in bf.knnMatch(d1, d2, k=2):
                if len(pr) == 2 and pr[0].distance < args.ratio * pr[1].distance:
                    p1, p2 = k1[pr[0].queryIdx].pt, k2[pr[0].trainIdx].pt
                    m.append((int(p1[0]), int(p1[1]), int(p2[0]), int(p2[1])))
        ref.append((feats[0], feats[1], m))
    dt = time.perf_counter() - t0

    def score(lists_of, feats_of):
        kp_same = desc_same = desc_all = set_same = list_same = 0
        jacc = []
        for i in range(n):
            ra, rb, rm = ref[i]
            for mine, theirs in zip(feats_of(i), (ra, rb)):
                kp_same += int(set(mine) == set(theirs))
                common = set(mine) & set(theirs)
                desc_all += len(common)
                desc_same += sum(1 for k_ in common if mine[k_] == theirs[k_])
            g = [tuple(r) for r in np.asarray(lists_of(i)).reshape(-1, 4).tolist()]
            list_same += int(g == rm)
            set_same += int(set(g) == set(rm))
            u = len(set(g) | set(rm))
            jacc.append(len(set(g) & set(rm)) / u if u else 1.0)
        return {"frames_with_identical_keypoint_set": kp_same, "of_frames": 2 * n,
                "descriptors_identical": desc_same, "of_common_keypoints": desc_all,
                "identical_match_sets": set_same, "identical_ordered_lists": list_same, "of_pairs": n,
                "mean_match_set_jaccard": round(float(np.mean(jacc)), 4)}

    sweep = []
    for combo in itertools.product((0, 1), repeat=len(ORB_SWITCHES)):
        kw = dict(zip(ORB_SWITCHES, combo))
        orc = oracle_orb.FeatureMatcherOracle(args.ratio, **kw)
        cache = {}

        def run(i, orc=orc, cache=cache):
            if i not in cache:
                (ka, da), (kb, db) = orc.extract_both(A[i], B[i])
                f = [{(int(q["octave"]), np.float32(q["x"]).tobytes(), np.float32(q["y"]).tobytes()): d_[j].tobytes()
                      for j, q in enumerate(k_)} for k_, d_ in ((ka, da), (kb, db))]
                cache[i] = (f, oracle_orb.knn_match(ka, da, kb, db, args.ratio))
            return cache[i]
        r = score(lambda i: run(i)[1], lambda i: run(i)[0])
        r["switches"] = kw
        sweep.append(r)
    rank = lambda r: (r["frames_with_identical_keypoint_set"], r["descriptors_identical"], r["identical_match_sets"])
    best = max(sweep, key=rank)
    default = next(r for r in sweep if r["switches"] == {"blur_tie_even": 1, "level_size_mul_inv": 0, "blur_kernel_sum256": 0})
    gpu = score(lambda i: gpu_lists[i], lambda i: ({}, {}))
    return {"opencv": cv2.__version__, "pairs": n, "pairs_per_sec_1thread": round(n / dt, 3),
            "gpu_default_vs_opencv": {k: gpu[k] for k in ("identical_match_sets", "identical_ordered_lists", "of_pairs",
                                                          "mean_match_set_jaccard")},
            "restatement_default_vs_opencv": default,
            "best_switches": best["switches"], "best_is_exact": bool(
                best["frames_with_identical_keypoint_set"] == 2 * n and best["descriptors_identical"] == best["of_common_keypoints"]
                and best["identical_match_sets"] == n),
            "default_is_best": rank(default) == rank(best), "sweep": sweep}


def two_handle_leg(args, matcher, fm, W, H, P, ratio_or_thr, local_rank, dev, dA, dB, out, cnt, packed, offs, side):
    """NOT the headline: what a caller's double buffering adds.  A second handle with buffers and a stream of its own takes
    every other step, so that the last round of workgroups of a step's kernels and its small tail kernels run beside the
    other step's walker / strips (one handle's launches are a dependent chain on one stream: every kernel boundary drains
    the chip).  Timed like the headline (same K, same barrier + synchronize bracket), in a region of its own after it;
    stage times are not taken here -- a stage's HIP-event interval would contain the other handle's kernels -- which is
    why `value`, `roofline` and the committed profiles stay with ONE handle."""
    import torch
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher
    if matcher == "orb":
        fm2 = FeatureMatcher(ratio_or_thr, W, H, device=local_rank, max_batch_pairs=P)
    else:
        fm2 = DNNFeatureMatcher(threshold=ratio_or_thr, device=local_rank, max_batch_pairs=P)
    out2, cnt2, packed2, offs2 = torch.zeros_like(out), torch.zeros_like(cnt), torch.zeros_like(packed), torch.zeros_like(offs)
    side2 = torch.cuda.Stream(device=dev)
    lanes = ((fm, out, cnt, packed, offs, side), (fm2, out2, cnt2, packed2, offs2, side2))

    def run(n):
        for k in range(n):
            h, o, c, pk, of, st = lanes[k & 1]
            h.match_batch_device(dA, dB, o, c, stream=st.cuda_stream)
            h.pack_matches_device(o, c, pk, of, stream=st.cuda_stream)

    run(max(2, args.warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = bool(torch.equal(cnt, cnt2) and torch.equal(offs, offs2) and torch.equal(packed, packed2))
    fm.stage_times()
    fm2.close()
    del out2, cnt2, packed2, offs2
    return {"value": round(P * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 4), "steps": args.steps,
            "handles": 2, "both_handles_identical_lists": same,
            "note": "steps alternate between two handles on two streams (the caller's double buffering; nothing in the "
                    "library); not the headline: `value` / `roofline` above are one handle, one stream"}


def run_workload(args, matcher, W, H, P, ratio_or_thr, loftr_f32, rank, world, local_rank, dev, cdev, with_cpu):
    """One metric configuration: P synthetic pairs per GPU resident in HBM, `warmup` untimed + `steps` timed steps on a
    stream of its own.  Returns the result fields of one JSON line (rank 0) or None."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from mono_slam_framework_amd import _lib, synth
    from mono_slam_framework_amd.gather import MatchListGather, RcclMatchListGather
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher

    mode = args.synth_mode if args.synth_mode is not None else (1 if matcher == "loftr" else 0)
    first = synth_first_pair(rank, world, P)
    A, B = synth.synth_batch(first, P, W, H, mode=mode, threads=min(16, os.cpu_count() or 1))
    dA, dB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    if matcher == "orb":
        fm = FeatureMatcher(ratio_or_thr, W, H, device=local_rank, max_batch_pairs=P, flags=_lib.MSF_FLAG_PROFILE)
    else:
        fm = DNNFeatureMatcher(threshold=ratio_or_thr, device=local_rank, max_batch_pairs=P,
                               flags=_lib.MSF_FLAG_PROFILE | (_lib.MSF_FLAG_LOFTR_F32 if loftr_f32 else 0))
    out = torch.zeros((P, args.cap, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros((P,), dtype=torch.int32, device=dev)
    # N > 1: the packed lists are double-buffered, so that the gather of step k (its own stream) runs beside the
    # kernels of step k + 1
    nbuf = 2 if world > 1 else 1
    packed_b = [torch.zeros((P * args.cap, 4), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    offs_b = [torch.zeros((P + 1,), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    packed, offs = packed_b[0], offs_b[0]
    torch.cuda.synchronize()
    # the whole step is enqueued on ONE stream of its own (the *_device entry points are asynchronous on the stream
    # they are given; one stream in flight per handle: include/msf_abi.h)
    side = torch.cuda.Stream(device=dev)
    stream = side.cuda_stream
    gather, gstream = None, None
    if world > 1:
        gstream = torch.cuda.Stream(device=dev)
        if args.gather == "product":
            # the product-side gather of libmsf.so (msf_gather_*: RCCL bound by the library); its 128-byte id travels
            # over the process group that exists anyway
            # RCCL refuses two ranks on one device: on a box with fewer GPUs than ranks the product gather can only be
            # REHEARSED, against the test-only stand-in named by MSF_RCCL_LIBRARY (tests/stub_rccl: host shared memory +
            # hipMemcpy instead of xGMI) -- such a line says so in config.gather and is no RCCL measurement
            if args.backend != "nccl" and not os.environ.get("MSF_RCCL_LIBRARY"):
                raise SystemExit("--gather product needs one GPU per rank (--backend nccl), or MSF_RCCL_LIBRARY naming the "
                                 "test stand-in for a rehearsal: RCCL refuses two ranks on one device")
            idt = torch.zeros((128,), dtype=torch.uint8, device=cdev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(RcclMatchListGather.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            gather = RcclMatchListGather(P, local_rank, rank, world, bytes(idt.cpu().numpy().tobytes()), P * args.cap)
        else:
            gather = MatchListGather(P, cdev, capacity_records=P * args.cap)
    ev_packed = [torch.cuda.Event() for _ in range(nbuf)]     # step's lists are packed (recorded on `side`)
    ev_gathered = [torch.cuda.Event() for _ in range(nbuf)]   # ... and have been sent (recorded on `gstream`)
    stage_acc = {}
    gathered = [0]

    def compute(k):
        """extract + match + pack of step k on the compute stream; nothing here waits on the host"""
        b = k % nbuf
        with torch.cuda.stream(side):
            if gather is not None and k >= nbuf:
                side.wait_event(ev_gathered[b])            # the gather of step k - 2 has read this buffer
            fm.match_batch_device(dA, dB, out, cnt, stream=stream)
            fm.pack_matches_device(out, cnt, packed_b[b], offs_b[b], stream=stream)
            ev_packed[b].record(side)

    def do_gather(k):
        """gather of step k's variable-length match lists to rank 0 (all-gather of offsets, then exact-size
        ncclSend / ncclRecv over xGMI; no all-reduce in the data path) on the gather stream: its one host wait -- the
        totals -- is a wait for THAT stream, the compute stream already holds the next step's kernels"""
        b = k % nbuf
        with torch.cuda.stream(gstream):
            gstream.wait_event(ev_packed[b])
            if args.gather == "product":
                res = gather(packed_b[b], offs_b[b], stream=gstream.cuda_stream)
            else:
                res = gather(packed_b[b].to(cdev), offs_b[b].to(cdev))
            ev_gathered[b].record(gstream)
        if res is not None:
            gathered[0] = sum(int(r[0].shape[0]) for r in res)

    def run_steps(n):
        if gather is None:
            for k in range(n):
                compute(k)
            return
        compute(0)
        for k in range(1, n):
            compute(k)               # step k's kernels are enqueued ...
            do_gather(k - 1)         # ... before the host waits for step k - 1's totals
        do_gather(n - 1)


    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    fence()
    fm.stage_times()          # drop the warm-up calls' stage times (msf_stage_times sums over the calls since the last query)
    t0 = time.perf_counter()
    run_steps(args.steps)     # N = 1: enqueue only, nothing waits for the device inside the timed region
    fence()
    dt = time.perf_counter() - t0
    # HIP events recorded on the launch stream by every timed call (a ring of event sets), read after the region
    for k, v in fm.stage_times().items():
        stage_acc[k] = stage_acc.get(k, 0.0) + v
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)   # timing only, outside the timed region
        dt = float(tmax.item())
    res = None
    if rank == 0:
        cnt_h = cnt.cpu().numpy()
        out_h = out.cpu().numpy()
        lists = [out_h[i, :max(cnt_h[i], 0)] for i in range(P)]
        ms_per_step = dt / args.steps * 1e3
        value = world * P * args.steps / dt
        bpp = algorithmic_bytes_per_pair(W, H)
        stages = {k: v / args.steps for k, v in stage_acc.items()}
        dom = max(stages, key=stages.get) if stages else None
        roofline = None
        traffic = traffic_record("loftr_f32" if loftr_f32 else matcher, dom, P, W, H)
        if dom and matcher == "loftr":
            # dominant "kernel" = the convolution stack of one call (peak = 157.3 TF where the f32 MFMA is
            # used, 2.5 PF where the operands are bf16 -- "state which")
            flops = P * (LOFTR_CONV_FLOPS_PER_PAIR if dom == "backbone_convs" else LOFTR_FLOPS_PER_PAIR)
            achieved = flops / (stages[dom] * 1e-3) / 1e12
            peak = MFMA_F32_PEAK_TFLOPS if loftr_f32 else MFMA_BF16_PEAK_TFLOPS
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 5), "traffic": traffic,
                        "avg_launch_ms": round(stages[dom], 4), "algorithmic_flops_per_launch": flops,
                        "stage_ms": {k: round(v, 4) for k, v in stages.items()},
                        "pipeline_frac": round(value / world * LOFTR_FLOPS_PER_PAIR / 1e12 / peak, 5),
                        "mfma_form": ("f32: v_mfma_f32_16x16x4_f32 everywhere, peak = the f32 matrix rate" if loftr_f32 else
                                      "split-bf16: each f32 product of the ResNet = 3 x v_mfma_f32_16x16x32_bf16 on hi/lo "
                                      "operands (f32 accumulation), peak = the dense bf16 matrix rate"),
                        "hbm_frac": hbm_frac_of(traffic, stages[dom])}
            if not loftr_f32:
                # algorithmic FLOPs x 3 = the matrix work actually issued on the split layers; the same time against the
                # f32 matrix peak is what r01/r02 printed as `frac` (no utilisation figure: the f32 pipe is not used)
                roofline["issued_frac"] = round(3 * achieved / peak, 5)
                roofline["f32_equiv_frac"] = round(achieved / MFMA_F32_PEAK_TFLOPS, 5)
            roofline["mfma_busy"] = mfma_busy_record(dom, P, loftr_f32)
        elif dom:
            # the dominant kernel k_walk (pyramid + FAST of all levels) is ONE launch per step: the stage time between
            # its two HIP events is that launch's duration (rocprofv3's average for k_walk must agree)
            launches = fm.walker_launches(dom)
            achieved = P * bpp / (stages[dom] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom,
                        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_ms": round(stages[dom] / launches, 4), "launches_per_step": launches,
                        "algorithmic_bytes_per_launch": P * bpp // launches,
                        "stage_ms": {k: round(v, 4) for k, v in stages.items()},
                        "pipeline_frac": round(value / world * bpp / 1e9 / HBM_PEAK_GBS, 5),
                        # the stage's measured HBM bytes (committed PMC passes) over its time: how busy the bus really is
                        "hbm_frac": hbm_frac_of(traffic, stages[dom])}
        if world > 1:
            assert gathered[0] > 0, "rank 0 gathered no match records"
        res = {
            "value": round(value, 2), "ms_per_step": round(ms_per_step, 4),
            "dtype": "u8" if matcher == "orb" else ("f32" if loftr_f32 else "f32 (split-bf16 MFMA products, f32 accumulation)"),
            "config": {"workload": "%s extract+match, %dx%d pairs, %d pairs/GPU/step resident in HBM, %s"
                                   % (matcher.upper(), W, H, P,
                                      "ratio %.2f" % ratio_or_thr if matcher == "orb" else "conf threshold %.2f" % ratio_or_thr),
                       "pairs_per_gpu": P, "width": W, "height": H, "synth_mode": mode,
                       "matches_per_pair_mean": round(float(np.mean([len(l) for l in lists])), 2),
                       "overflow_pairs": int((cnt_h < 0).sum()),
                       "gathered_match_records_per_step": gathered[0] if world > 1 else int(offs[P].item()),
                       "shard": "pair p of a step -> rank p // pairs_per_gpu (contiguous blocks)",
                       "collective_backend": args.backend if world > 1 else None,
                       "gather": (None if world == 1 else
                                  ("product: msf_gather_* of libmsf.so (ncclAllGather + ncclSend/ncclRecv)" +
                                   (" bound to the stand-in MSF_RCCL_LIBRARY=%s: a REHEARSAL of offsets and ordering, "
                                    "not RCCL" % os.path.basename(os.environ["MSF_RCCL_LIBRARY"])
                                    if os.environ.get("MSF_RCCL_LIBRARY") else "") if args.gather == "product"
                                   else "torch.distributed (all_gather_into_tensor + batch_isend_irecv)") +
                                  ", pipelined: step k's gather on its own stream beside step k+1's kernels")},
            "roofline": roofline,
        }
        if world == 1 and not args.no_two_handles and not loftr_f32:
            try:
                res["two_handles"] = two_handle_leg(args, matcher, fm, W, H, P, ratio_or_thr, local_rank, dev, dA, dB, out,
                                                    cnt, packed, offs, side)
            except Exception as e:      # an extra record: never at the price of the line
                res["two_handles"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if with_cpu:
            cargs = argparse.Namespace(**vars(args))
            cargs.width, cargs.height, cargs.ratio, cargs.threshold = W, H, ratio_or_thr, ratio_or_thr
            res["cpu_baseline"] = (cpu_baseline if matcher == "orb" else cpu_baseline_loftr)(cargs, A, B, lists)
            if matcher == "orb":
                res["cpu_baseline"]["reference_lib"] = opencv_probe(cargs, A, B, lists)
    fm.close()
    if world > 1 and args.gather == "product":
        gather.close()
    del dA, dB, out, cnt, packed, offs, packed_b, offs_b
    torch.cuda.empty_cache()
    return res


HBM_FRAC_PLAUSIBLE = 0.85   # measured streaming copies reach 0.79 of the 8 TB/s spec (MI355X_MICROARCH.md, HBM)


def hbm_frac_of(traffic, stage_ms):
    """Measured HBM bytes of the stage (committed PMC pass) over its time, as a fraction of the 8 TB/s peak.  A record
    that would put the bus above what a pure copy reaches is a broken record, not a fast kernel (r04's were 3 x one
    pass): it is marked `implausible` and no fraction is printed for it."""
    if not traffic:
        return None
    frac = traffic["bytes"] / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if frac > HBM_FRAC_PLAUSIBLE:
        traffic["implausible"] = True
        traffic["implied_hbm_frac"] = round(frac, 5)
        return None
    return round(frac, 5)


def synth_first_pair(rank, world, P):
    # contiguous blocks: rank r owns pairs r*P .. r*P + P - 1 of every step (gather.shard_pairs)
    return rank * P


def csrc_digest():
    """sha256 (16 hex digits) over the kernel and host sources of libmsf.so: what a committed counter pass was measured on"""
    import hashlib
    d = os.path.join(ROOT, "mono_slam_framework_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cpp", ".h", ".c")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def traffic_record(matcher, dom, P, W, H):
    """HBM bytes per launch of the dominant stage from the committed PMC passes (profiles/traffic_*.json, collected by
    tools/collect_profiles.sh: separate --pmc runs, FETCH_SIZE x 2 + WRITE_SIZE).  Not measured inside this run: the
    record names its source and the kernel versions it was taken at, and is dropped when the batch differs."""
    name = "orb_vga" if (matcher == "orb" and W == 640) else matcher
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % name)
    try:
        tj = json.load(open(tfile))
    except Exception:
        return None
    if tj.get("_pairs_per_gpu") != P or tj.get("_width", W) != W or dom not in tj:
        return None
    return {"bytes": tj[dom], "per": "step (all launches of the stage)", "source": "profiles/traffic_%s.json" % name,
            "measured_at": tj.get("_commit"),
            # the kernel sources have changed since the counter pass: the figure describes an older build
            "stale": tj.get("_csrc_sha16") != csrc_digest()}


def mfma_busy_record(dom, P, loftr_f32):
    """Matrix-pipe utilisation of the dominant stage from the committed counter pass (profiles/traffic_loftr*.json,
    tools/make_traffic_json.py): sum over the stage's dispatches of SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs
    x 1024 SIMDs), i.e. the fraction of SIMD-cycles in which a matrix instruction executes.  Not measured inside this
    run (PMC needs rocprofv3); None when no pass is committed for this variant / batch."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_loftr%s.json" % ("_f32" if loftr_f32 else ""))))
    except Exception:
        return None
    mb = tj.get("_mfma_busy")
    if not mb or tj.get("_pairs_per_gpu") != P or dom not in mb:
        return None
    return {"value": mb[dom], "per_kernel": mb.get("_per_kernel"), "derivation": mb.get("_derivation"),
            "source": "profiles/traffic_loftr%s.json" % ("_f32" if loftr_f32 else ""), "measured_at": tj.get("_commit"),
            "stale": tj.get("_csrc_sha16") != csrc_digest()}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started as a plain process: start the N ranks as a CHILD torch.distributed.run
    and relay its output -- before anything here has touched the GPU (torch is not even imported in this process)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU per step (resident in HBM); default 1024 (ORB 720p), 4096 (ORB VGA), 256 (LoFTR)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--ratio", type=float, default=0.6, help="Lowe ratio (the app uses 0.6, src/main.cpp:66)")
    ap.add_argument("--cap", type=int, default=1024, help="match-list capacity per pair")
    ap.add_argument("--matcher", default=None, choices=["orb", "loftr"],
                    help="one workload only; default: the headline (ORB 1280x720) plus, on one GPU, the two other metric "
                         "configurations (ORB 640x480, LoFTR 640x480) as `secondary`")
    ap.add_argument("--synth-mode", type=int, default=None,
                    help="synthetic texture: 0 blocky x8 (SURVEY 8d, ORB default), 1 smooth blobs (LoFTR default), 2 blocky x16")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-pairs-per-thread", type=int, default=2)
    ap.add_argument("--cpu-loftr-pairs", type=int, default=8)
    ap.add_argument("--cpu-extra-seconds", type=float, default=5.0,
                    help="wall time of each of the two further CPU legs (single thread, all usable hardware threads); 0 = skip")
    ap.add_argument("--cpu-seconds", type=float, default=8.0,
                    help="wall time each CPU-baseline leg aims for (its sample is sized from a one-pair calibration)")
    ap.add_argument("--threshold", type=float, default=0.15, help="LoFTR confidence threshold (dnnfeaturematcher.h:11)")
    ap.add_argument("--loftr-f32", action="store_true",
                    help="LoFTR: MSF_FLAG_LOFTR_F32 (every convolution on the f32 MFMA instead of split-bf16 products)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-two-handles", action="store_true",
                    help="skip the extra `two_handles` record (steps alternating between two handles on two streams)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo = rehearsal of the N > 1 code path "
                         "on a box with fewer GPUs than ranks (all ranks share cuda:0, results staged through host)")
    ap.add_argument("--gather", default="torch", choices=["torch", "product"],
                    help="N > 1: the gather of the match lists to rank 0 through torch.distributed (default) or through "
                         "the product's own msf_gather_* (libmsf.so binds RCCL itself; never run on more than one GPU yet)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): one rank per GPU" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the collective's tensors live

    with_cpu = not args.no_cpu_baseline and world == 1     # the CPU leg is timed on rank 0 of the 1-GPU run only
    if args.matcher == "loftr":
        head = ("loftr", 640, 480, args.pairs or 256, args.threshold, args.loftr_f32)
    else:
        vga = (args.width, args.height) == (640, 480)
        head = ("orb", args.width, args.height, args.pairs or (4096 if vga else 1024), args.ratio, False)
    extra = []
    if args.matcher is None and world == 1 and not args.no_secondary and (args.width, args.height) == (1280, 720) \
            and args.pairs is None:
        extra = [("orb", 640, 480, 4096, args.ratio, False), ("loftr", 640, 480, 256, args.threshold, False),
                 ("loftr", 640, 480, 256, args.threshold, True)]     # the exact-f32 LoFTR path, driver-timed too
    devices = [(local_rank, torch.cuda.get_device_name(local_rank))]
    if world > 1:
        gathered_dev = [None] * world
        dist.all_gather_object(gathered_dev, devices[0])
        devices = gathered_dev
    r = run_workload(args, *head, rank, world, local_rank, dev, cdev, with_cpu)
    secondary = [run_workload(args, *w, rank, world, local_rank, dev, cdev, with_cpu and not w[5]) for w in extra]
    if rank == 0:
        r["config"].update({"ranks": world, "rccl_ranks": world if (world > 1 and args.backend == "nccl") else 0,
                            "collective_backend": (args.backend if world > 1 else None),
                            "rank_devices": ["rank %d: cuda:%d %s" % (i, d[0], d[1]) for i, d in enumerate(devices)]})
        line = {"metric": "frame-pairs/sec (extract+match)", "value": r["value"], "unit": "frame-pairs/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": r["dtype"],
                "data": "synthetic", "config": r["config"], "roofline": r["roofline"]}
        if "cpu_baseline" in r:
            line["cpu_baseline"] = r["cpu_baseline"]
        if "two_handles" in r:
            line["two_handles"] = r["two_handles"]
        if secondary:
            # the two other configurations of BASELINE.json's metric, each run like the headline: same steps / warmup
            line["secondary"] = [dict(workload=x["config"]["workload"], unit="frame-pairs/sec", steps=args.steps,
                                      warmup=args.warmup, **{k: v for k, v in x.items() if k != "config"},
                                      config=x["config"]) for x in secondary]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
