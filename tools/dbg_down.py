"""Debug aid: output of layer2's first block from k_down16x against a numpy evaluation on the dumped layer-1 activation."""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbg_split as D
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blob(path):
    b = open(path, "rb").read()
    assert b[:8] == b"MSFLTR01"
    n = struct.unpack_from("<I", b, 8)[0]
    recs, off = {}, 12
    for _ in range(n):
        name = b[off:off + 32].split(b"\0")[0].decode()
        ndim, d0, d1, d2, d3, fo, cnt = struct.unpack_from("<7I", b, off + 32)
        recs[name] = (fo, cnt, (d0, d1, d2, d3)[:ndim])
        off += 32 + 28
    out = {}
    for k, (fo, cnt, dims) in recs.items():
        out[k] = np.frombuffer(b, np.float32, cnt, off + 4 * fo).reshape(dims)
    return out


def conv(x, w, b, stride, pad):
    co, ci, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad)))
    H = (x.shape[1] + 2 * pad - kh) // stride + 1
    W = (x.shape[2] + 2 * pad - kw) // stride + 1
    y = np.zeros((co, H, W), np.float64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + stride * H:stride, kx:kx + stride * W:stride].astype(np.float64)
            y += np.einsum("oc,chw->ohw", w[:, :, ky, kx].astype(np.float64), patch)
    return (y + b[:, None, None]).astype(np.float32)


os.environ["MSF_DBG_ACT1_CC"] = "1"
x = D.run({"MSF_LOFTR_STRIP_MIN": "1"})
W = blob(os.path.join(ROOT, "mono_slam_framework_amd", "weights", "loftr_teacher.bin"))
a0 = x["a0"]
t = np.maximum(conv(a0, W["conv05.w"], W["conv05.b"], 2, 1), 0)
sc = conv(a0, W["conv07.w"], W["conv07.b"], 2, 0)
u = np.maximum(conv(t, W["conv06.w"], W["conv06.b"], 1, 1) + sc, 0)
u_nosc = np.maximum(conv(t, W["conv06.w"], W["conv06.b"], 1, 1), 0)
u_sconly = np.maximum(sc, 0)
for mode, ref, nm in ((1, t, "t"), (2, sc, "sc")):
    xd = D.run({"MSF_LOFTR_STRIP_MIN": "1", "MSF_DBG_DOWN": str(mode)})["a1"]
    d = np.abs(xd - ref)
    print("%s: max |gpu - ref| %.4g median %.3g; per channel max %s" % (nm, d.max(), np.median(d), np.round(d.max(axis=(1, 2)), 2)))
    pm = d.max(axis=0)
    print("   by row mod 4:", np.round(pm.reshape(30, 4, 160).max(axis=(0, 2)), 3), " by col mod 16:", np.round(pm.reshape(120, 10, 16).max(axis=(0, 1)), 2))
    print("   gpu", np.round(xd[0, 40, 40:46], 3), "ref", np.round(ref[0, 40, 40:46], 3), "ref(row+1)", np.round(ref[0, 41, 40:46], 3), "ref(col+1)", np.round(ref[0, 40, 41:47], 3))
w5, b5 = W["conv05.w"], W["conv05.b"]
for ky in range(0):
    for kx in range(3):
        xd = D.run({"MSF_LOFTR_STRIP_MIN": "1", "MSF_DBG_DOWN": "1", "MSF_DBG_W1": "%d,%d" % (ky, kx)})["a1"]
        m = np.zeros_like(w5); m[:, :, ky, kx] = w5[:, :, ky, kx]
        r = np.maximum(conv(a0, m, b5, 2, 1), 0)
        # try shifted alignments of the reference to see where the tap really reads
        best = min(((np.abs(xd[:, 4:-4, 4:-4] - np.roll(r, (dy, dx), (1, 2))[:, 4:-4, 4:-4]).max(), dy, dx) for dy in (-1, 0, 1) for dx in (-2, -1, 0, 1, 2)))
        print("tap (%d,%d): max diff %.4g; best shift (dy,dx)=(%d,%d) diff %.4g" % (ky, kx, np.abs(xd - r).max(), best[1], best[2], best[0]))
