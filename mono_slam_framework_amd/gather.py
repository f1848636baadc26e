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
