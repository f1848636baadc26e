"""What would layer 1 (stem + 4 convs) with f16-rounded weights cost in confidence?  CPU only: the f32 C restatement run
with a weights blob whose layer-1 tensors are rounded to f16 (what a two-product x_hi*w + x_lo*w would compute)."""
import struct, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loftr as ol
from mono_slam_framework_amd import synth
src = ol.WEIGHTS
b = open(src, 'rb').read()
n = struct.unpack_from('<I', b, 8)[0]
recs = []
for i in range(n):
    o = 12 + 60 * i
    name = b[o:o+32].split(b'\0')[0].decode()
    ndim, d0, d1, d2, d3, off, cnt = struct.unpack_from('<7I', b, o + 32)
    recs.append((name, ndim, (d0, d1, d2, d3)[:ndim], off, cnt))
payload = 12 + 60 * n
data = np.frombuffer(b, dtype='<f4', offset=payload).copy()
names = [r[0] for r in recs]
print(len(recs), names[:12])
def variant(which, fmt):
    d = data.copy()
    for (name, ndim, dims, off, cnt) in recs:
        if which(name, dims):
            w = d[off:off+cnt]
            if fmt == 'f16':
                d[off:off+cnt] = w.astype(np.float16).astype(np.float32)
            elif fmt == 'bf16':
                u = w.view(np.uint32); r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32); d[off:off+cnt] = r.view(np.float32)
    out = '/tmp/w_%s.bin' % fmt
    open(out, 'wb').write(b[:payload] + d.astype('<f4').tobytes())
    return out
# layer-1 tensors: conv weights with cout = 8 (stem 8x1x7x7, four 8x8x3x3)
is_l1 = lambda name, dims: len(dims) == 4 and dims[0] == 8
print([r[0] for r in recs if is_l1(r[0], r[2])])
A, B = synth.synth_batch(777, 6, 640, 480, mode=1)
pairs = [(A[i], B[i]) for i in range(6)] + [(synth.kat_pattern(640, 480, 0, 0), synth.kat_pattern(640, 480, 32, 16))]
ref = ol.DNNFeatureMatcherOracle(0.15)
confs = [ref.run(a, c)['conf'] for a, c in pairs]
for fmt in ('f16', 'bf16'):
    orc = ol.DNNFeatureMatcherOracle(0.15, weights=variant(is_l1, fmt))
    worst = 0
    for (a, c), c0 in zip(pairs, confs):
        worst = max(worst, float(np.abs(orc.run(a, c)['conf'] - c0).max()))
    print('layer-1 weights rounded to', fmt, ': max |dconf| over', len(pairs), 'pairs =', worst)
