"""Multi-GPU result path: pairs shard over ranks with no data-path collective; the only exchange is
the gather of variable-length match lists to rank 0 -- an all-gather of the per-pair offsets, then exact-size
point-to-point sends (ncclSend/ncclRecv over xGMI under RCCL; gloo on CPU in the tests).  Never an all-reduce."""
import torch
import torch.distributed as dist


def shard_pairs(n_pairs_total, rank, world):
    """Static, deterministic partition into contiguous blocks (what bench.py and DESIGN.md section 6 use: a rank's
    frames are one resident array, and rank 0's gathered lists come out in pair order): rank r owns pairs
    [r * ceil(n / world), min(n, (r + 1) * ceil(n / world)))."""
    per = -(-n_pairs_total // world)
    return list(range(min(n_pairs_total, rank * per), min(n_pairs_total, (rank + 1) * per)))


class MatchListGather:
    def __init__(self, pairs_per_rank, device, group=None, capacity_records=0):
        """capacity_records > 0: rank 0's receive buffers are allocated here, once, for that many match records per
        rank (a step that stays within it allocates nothing); 0: they grow on demand."""
        self.P = pairs_per_rank
        self.dev = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.all_offs = torch.zeros((self.world * (self.P + 1),), dtype=torch.int32, device=device)
        self.recv = [None] * self.world
        if capacity_records > 0 and self.rank == 0:
            for r in range(1, self.world):
                self.recv[r] = torch.empty((capacity_records, 4), dtype=torch.int32, device=device)

    def __call__(self, packed, offsets):
        """packed int32 [>= total, 4], offsets int32 [P + 1] (offsets[P] = total), both on self.dev.
        Returns on rank 0: list over ranks of (packed_r [total_r, 4], offsets_r [P + 1]); elsewhere None."""
        if self.world == 1:
            return [(packed[:int(offsets[self.P])], offsets)]
        dist.all_gather_into_tensor(self.all_offs, offsets.contiguous(), group=self.group)
        offs = self.all_offs.view(self.world, self.P + 1)
        # one small D2H read per step: exact-size send / recv need the totals on the host (the alternative, a padded
        # all-gather of capacity-sized lists, moves cap / mean-length = ~10x the bytes over xGMI)
        totals = offs[:, self.P].tolist()
        ops = []
        if self.rank == 0:
            for r in range(1, self.world):
                if self.recv[r] is None or self.recv[r].shape[0] < totals[r]:
                    self.recv[r] = torch.empty((max(totals[r], 1) * 2, 4), dtype=torch.int32, device=self.dev)
                if totals[r]:
                    ops.append(dist.P2POp(dist.irecv, self.recv[r][:totals[r]], r, group=self.group))
        elif totals[self.rank]:
            ops.append(dist.P2POp(dist.isend, packed[:totals[self.rank]], 0, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.rank != 0:
            return None
        out = [(packed[:totals[0]], offs[0])]
        for r in range(1, self.world):
            out.append((self.recv[r][:totals[r]] if totals[r] else packed[:0], offs[r]))
        return out


def gather_plan(all_offsets, cap_records):
    """The placement arithmetic of the product-side gather (msf_gather_plan, csrc/msf_gather.cpp) on host arrays:
    all_offsets int32 [n_ranks, P + 1] -> (status, totals int32 [n_ranks], recv_first int64 [n_ranks])."""
    import numpy as np
    from . import _lib
    L = _lib.load()
    o = np.ascontiguousarray(all_offsets, np.int32)
    totals = np.zeros((o.shape[0],), np.int32)
    first = np.zeros((o.shape[0],), np.int64)
    rc = L.msf_gather_plan(o.shape[0], o.shape[1] - 1, o.ctypes.data, int(cap_records), totals.ctypes.data, first.ctypes.data)
    return rc, totals, first


class RcclMatchListGather:
    """The product-side form of MatchListGather: msf_gather_* of libmsf.so (RCCL bound by the library itself, no
    torch.distributed involved).  The 128-byte id comes from rank 0 (unique_id()) through the job's own channel."""

    def __init__(self, pairs_per_rank, device, rank, n_ranks, id128, capacity_records):
        import ctypes as C
        import torch
        from . import _lib
        self._L = _lib.load()
        self.P, self.rank, self.world, self.cap = pairs_per_rank, rank, n_ranks, capacity_records
        self._g = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        rc = self._L.msf_gather_create(device, rank, n_ranks, buf, pairs_per_rank, capacity_records, C.byref(self._g))
        if rc != _lib.MSF_OK:
            raise RuntimeError("msf_gather_create: %s" % self._L.msf_gather_last_error(None).decode())
        dev = torch.device("cuda", device)
        self.all_offs = torch.zeros((n_ranks, pairs_per_rank + 1), dtype=torch.int32, device=dev)
        self.recv = torch.zeros((n_ranks * capacity_records, 4), dtype=torch.int32, device=dev) if rank == 0 else None

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        L = _lib.load()
        buf = (C.c_uint8 * 128)()
        if L.msf_gather_unique_id(buf) != _lib.MSF_OK:
            raise RuntimeError("msf_gather_unique_id: %s" % L.msf_gather_last_error(None).decode())
        return bytes(buf)

    def __call__(self, packed, offsets, stream=None):
        """as MatchListGather.__call__: on rank 0 a list over ranks of (records [total_r, 4], offsets_r [P + 1])"""
        import numpy as np
        import torch
        from . import _lib
        totals = np.zeros((self.world,), np.int32)
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        rc = self._L.msf_gather_matches_device(self._g, packed.data_ptr(), offsets.data_ptr(), self.all_offs.data_ptr(),
                                               self.recv.data_ptr() if self.recv is not None else None,
                                               totals.ctypes.data, st)
        if rc != _lib.MSF_OK:
            raise RuntimeError("msf_gather_matches_device: %s" % self._L.msf_gather_last_error(self._g).decode())
        if self.rank != 0:
            return None
        out, at = [], 0
        for r in range(self.world):
            out.append((self.recv[at:at + int(totals[r])], self.all_offs[r]))
            at += int(totals[r])
        return out

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._L.msf_gather_destroy(self._g)
            self._g.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
