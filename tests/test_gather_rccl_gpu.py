"""GPU (one MI355X): the product-side gather msf_gather_* (csrc/msf_gather.cpp) with a communicator of ONE rank -- RCCL
is found and bound, ncclCommInitRank, the ncclAllGather of the offsets, the totals on the host and rank 0's own records
run on real hardware and reproduce what msf_pack_matches_device produced.  The ncclSend / ncclRecv leg needs a second GPU
and has not executed anywhere yet."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_round_trip():
    import torch
    from mono_slam_framework_amd import synth
    from mono_slam_framework_amd.gather import RcclMatchListGather
    from mono_slam_framework_amd.matcher import FeatureMatcher
    n, w, h, cap = 6, 320, 240, 1024
    A, B = synth.synth_batch(4200, n, w, h)
    dev = torch.device("cuda", 0)
    fm = FeatureMatcher(0.7, w, h, max_batch_pairs=n)
    dA, dB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    out = torch.zeros((n, cap, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
    packed = torch.zeros((n * cap, 4), dtype=torch.int32, device=dev)
    offs = torch.zeros((n + 1,), dtype=torch.int32, device=dev)
    fm.match_batch_device(dA, dB, out, cnt)
    fm.pack_matches_device(out, cnt, packed, offs)
    torch.cuda.synchronize()
    g = RcclMatchListGather(n, 0, 0, 1, RcclMatchListGather.unique_id(), n * cap)
    res = g(packed, offs)
    torch.cuda.synchronize()
    assert len(res) == 1
    rec, o = res[0][0].cpu().numpy(), res[0][1].cpu().numpy()
    np.testing.assert_array_equal(o, offs.cpu().numpy())
    total = int(o[-1])
    assert total > 50 and rec.shape == (total, 4)
    np.testing.assert_array_equal(rec, packed[:total].cpu().numpy())
    c = cnt.cpu().numpy()
    for i in range(n):
        np.testing.assert_array_equal(rec[o[i]:o[i + 1]], out[i, :c[i]].cpu().numpy())
    # too small a capacity is reported, not overrun
    g2 = RcclMatchListGather(n, 0, 0, 1, RcclMatchListGather.unique_id(), 8)
    with pytest.raises(RuntimeError, match="cap_records"):
        g2(packed, offs)
    g.close()
    g2.close()
