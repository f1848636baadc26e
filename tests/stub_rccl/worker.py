"""One rank of tests/test_gather_stub_gpu.py (TEST INFRASTRUCTURE): drives msf_gather_* of libmsf.so through the
scenarios below with MSF_RCCL_LIBRARY pointing at tests/stub_rccl/libstub_rccl.so, all ranks sharing cuda:0.
argv: rank n_ranks id_file.  Prints one JSON object (per scenario: return code, totals, and on rank 0 whether every
received record sits where msf_gather_plan says)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

P, CAP = 5, 400_000          # pairs per rank, record capacity per rank (6.4 MB: more than one 4-MB stub chunk)


def rank_data(rank, scenario, total):
    """deterministic lists of one rank: per-pair counts (sum = total) and records [rank, scenario, i, 7 i]"""
    rng = np.random.default_rng(1000 * scenario + rank)
    cuts = np.sort(rng.integers(0, total + 1, size=P - 1)) if total else np.zeros(P - 1, np.int64)
    offs = np.concatenate([[0], cuts, [total]]).astype(np.int32)
    i = np.arange(total, dtype=np.int32)
    rec = np.stack([np.full(total, rank, np.int32), np.full(total, scenario, np.int32), i, i * 7], axis=1) if total else np.zeros((0, 4), np.int32)
    return offs, rec


# scenario -> total records of (rank 0, every other rank)
SCENARIOS = [("normal", 37, 29), ("empty_rank", 11, 0), ("two_chunks", 5, 300_000), ("exact_capacity", 3, CAP),
             ("over_capacity", 3, CAP + 1), ("normal_again", 8, 13), ("injected_send_failure", 4, 6), ("after_failure", 4, 6)]


def main():
    rank, n, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch
    from mono_slam_framework_amd import _lib
    L = _lib.load()
    dev = torch.device("cuda", 0)
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        assert L.msf_gather_unique_id(buf) == 0, L.msf_gather_last_error(None)
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 60
            time.sleep(0.01)
    id128 = open(idfile, "rb").read()
    g = C.c_void_p()
    rc = L.msf_gather_create(0, rank, n, (C.c_uint8 * 128).from_buffer_copy(id128), P, CAP, C.byref(g))
    assert rc == 0, L.msf_gather_last_error(None)
    stream = torch.cuda.Stream(device=dev)
    all_offs = torch.zeros((n, P + 1), dtype=torch.int32, device=dev)
    recv = torch.full((n * CAP, 4), -1, dtype=torch.int32, device=dev) if rank == 0 else None
    packed = torch.zeros((CAP + 8, 4), dtype=torch.int32, device=dev)
    offs_d = torch.zeros((P + 1,), dtype=torch.int32, device=dev)
    out = {"rank": rank, "scenarios": {}}
    for si, (name, t0_, tr_) in enumerate(SCENARIOS):
        total = t0_ if rank == 0 else tr_
        offs, rec = rank_data(rank, si, total)
        h_rec = torch.from_numpy(np.ascontiguousarray(rec)).pin_memory() if total else None
        h_off = torch.from_numpy(offs).pin_memory()
        if recv is not None:
            recv.fill_(-1)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            # the lists are produced on the SAME stream right in front of the call, nothing waits on the host
            if total:
                packed[:total].copy_(h_rec, non_blocking=True)
            offs_d.copy_(h_off, non_blocking=True)
            totals = np.full((n,), -7, np.int32)
            rc = L.msf_gather_matches_device(g, packed.data_ptr(), offs_d.data_ptr(), all_offs.data_ptr(),
                                             recv.data_ptr() if recv is not None else None, totals.ctypes.data, stream.cuda_stream)
            packed.zero_()          # later work on the stream must not reach the bytes the gather sends
        stream.synchronize()
        res = {"rc": rc, "totals": totals.tolist(), "err": L.msf_gather_last_error(g).decode() if rc else ""}
        if rc == 0:
            ao = all_offs.cpu().numpy()
            exp_off = [rank_data(r, si, t0_ if r == 0 else tr_)[0] for r in range(n)]
            res["offsets_ok"] = bool(all(np.array_equal(ao[r], exp_off[r]) for r in range(n)))
            if rank == 0:
                got = recv.cpu().numpy()
                ok, at = True, 0
                for r in range(n):
                    o, e = rank_data(r, si, t0_ if r == 0 else tr_)
                    ok = ok and np.array_equal(got[at:at + len(e)], e)
                    # pair p of rank r starts at sum(totals[0..r)) + offsets[r][p]  (include/msf_abi.h)
                    for p in range(P):
                        ok = ok and np.array_equal(got[at + o[p]:at + o[p + 1]], e[o[p]:o[p + 1]])
                    at += len(e)
                ok = ok and bool((got[at:at + 16] == -1).all())          # nothing written past the last record
                res["records_ok"] = bool(ok)
        out["scenarios"][name] = res
    L.msf_gather_destroy(g)          # safe after the abort of the injected failure
    out["destroyed"] = True
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
