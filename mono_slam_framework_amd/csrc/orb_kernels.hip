// ORB extract + match kernels for gfx950 (CDNA4, wave64).
//
// What is computed (and the upstream routine each stage stands in for) is stated in
// DESIGN.md "ORB path"; the reference entry point is ::FeatureMatcher::MatchFrames
// (src/featurematcher.cpp:10-45), whose arithmetic is cv::ORB::detectAndCompute
// (cv::ORB::create() defaults) and BFMatcher(NORM_HAMMING)::knnMatch(k=2).
//
// Integer/byte work, HBM- and VALU-bound: no MFMA here.  This TU is compiled with
// -ffp-contract=off so the few f32 steps (Harris response, fastAtan2, pattern rotation)
// round exactly as the CPU restatement does.
#include <type_traits>
#include "orb_pipeline.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orb_pattern.h"

namespace msf {

// ------------------------------------------------------------------ constants
constexpr int kEdge = 31;          // edgeThreshold
constexpr int kFastT = 20;         // fastThreshold
constexpr int TW = 128, TH = 32;    // FAST output tile
constexpr int SW = TW + 2, SH = TH + 2;    // scored region: tile + 1 halo (NMS neighbours)
constexpr int kTileCandCap = TW * TH / 4;    // strict 3x3 maxima: at most one per 2x2
constexpr int kTileX0 = 16, kTileY0 = kEdge;  // tile grid origin: first output column 31 rounded down to the 16-byte load grid

constexpr uint32_t kStatusOverflow = 1u;

__constant__ __attribute__((aligned(16))) signed char c_pattern[1024];
// ICAngles disc as 31 rows x 8 dwords of 4 pixels (u = -16 + 4k .. -13 + 4k): per dword the signed byte weights u
// (0 outside the disc) and the signed byte weights v (the row's, 0 outside the disc), padded to 4 x 64 tasks
constexpr int kDiscTasks = 256;
__constant__ uint32_t c_disc[2 * kDiscTasks];

// ------------------------------------------------------------------ helpers
// a pointer / integer the caller knows to be the same in every lane, pinned to scalar registers.  A buffer descriptor
// built from such values gives loads and stores of the form buffer_load v, v_off, s[desc], s_off: the row offset is SALU
// work and no vector instruction is spent on addresses (the walker's VALU pipe is its bound); accesses past `bytes` are
// dropped by the hardware range check.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, uint32_t bytes) {
  const unsigned long long v = (unsigned long long)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ const uint8_t* level_ptr(const OrbGeometry& g, const FrameSrc& src,
                                                    const uint8_t* pyr, int fi, int l, int* pitch) {
  if (l == 0) {
    *pitch = src.row_stride;
    return fi < src.n_a ? src.a + (long long)fi * src.frame_stride
                        : src.b + (long long)(fi - src.n_a) * src.frame_stride;
  }
  *pitch = g.lv[l].pitch;
  return pyr + (long long)(src.slot0 + fi) * g.pyr_bytes + g.lv[l].pix_off;
}

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
// v_dot2_u32_u16: a.lo * b.lo + a.hi * b.hi + c on u16 pairs packed in dwords
__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b), c, false);
}

// v_mad_u32_u24: a * b + c on the low 24 bits of a and b (the compiler prefers two multiplies and an add3)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// the same with a wave-uniform multiplier taken from a scalar register (no v_mov to get it into a vector register)
__device__ __forceinline__ uint32_t mad_u24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}

// ------------------------------------------------------------------ K2: pyramid level l from l-1
// cv::resize(..., INTER_LINEAR_EXACT) restated: 8.8 fixed-point taps from host tables, 16-bit horizontal sums,
// 32-bit vertical, (v + 32768) >> 16.  A workgroup makes a band of `rth` (8) full output rows: the source rows it needs
// are staged whole in LDS with 16-byte loads (every fetched cache line is used once; 64-px-wide windows measured
// 2.7x over-fetch), then each lane blends 4 consecutive output pixels per task and stores them as one dword.
// LDS is read as ALIGNED dwords and the two taps are cut out with v_alignbyte: adjacent byte reads get fused by
// the compiler into misaligned ds_read_u16, which the LDS replays (measured: 5x slower kernel).
// Taps with weight 0 may read one byte past the image: staged as 0 (or padding), times 0.
// The kernel is VALU-bound, not HBM-bound, so the per-column work (tap offset, alignbyte shift, weight pair) is done
// once per task of 4 px x 4 rows instead of once per output dword, the per-row work comes from a small LDS table, and
// task / row decoding uses host-computed reciprocals instead of integer division (25 -> 12 VALU instructions per px).
constexpr int kResizeMaxRows = 16;     // output rows per band (rth) upper bound
// Per group of 4 output pixels the host table holds, ready for use: the byte offsets of the aligned dword pairs the taps
// are read from, the v_perm selectors that cut the two tap bytes out of a pair and widen them to u16 (ONE v_perm_b32
// instead of alignbyte + perm), and the weight pairs for v_dot2.
// SHARED: the taps of output pixels 4g, 4g+1, 4g+2 all lie inside the pair that holds the first one's (true for every
// level of a 1.2x pyramid; the host checks the table and falls back to per-pixel pairs otherwise), so one
// ds_read2_b32 serves three pixels.
template <bool SHARED>
__global__ __launch_bounds__(256) void k_resize(OrbGeometry g, FrameSrc src, uint8_t* pyr,
                                                const uint32_t* __restrict__ tab, int l, int rth, int lds_rows,
                                                uint32_t magic_n16, uint32_t magic_groups) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* ysh = reinterpret_cast<uint32_t*>(smem);   // [kResizeMaxRows] per output row: LDS byte offset, weight | reuse flag
  uint8_t* rt = smem + 4 * kResizeMaxRows * 2;
  const int fi = blockIdx.y;
  const OrbLevelInfo L = g.lv[l];
  const int sh = g.lv[l - 1].h, sw16 = (g.lv[l - 1].w + 16 + 15) & ~15;   // staged row: source width + 16 zero/pad bytes
  int spitch;
  const uint8_t* s = level_ptr(g, src, pyr, fi, l - 1, &spitch);
  uint8_t* d = pyr + (long long)(src.slot0 + fi) * g.pyr_bytes + L.pix_off;
  const int groups = (L.w + 3) >> 2;
  const uint4* xsel = reinterpret_cast<const uint4*>(tab + L.tab_off);   // [groups] selectors, weight pairs, pair offsets
  const uint4* xwxp = xsel + groups;
  const uint4* xoff = xwxp + groups;
  const uint32_t* ytab = reinterpret_cast<const uint32_t*>(xoff + groups);   // per y: source row | w1 << 16
  const int Y0 = blockIdx.x * rth, tid = threadIdx.x, nthr = blockDim.x;
  const int ylast = min(Y0 + rth, L.h) - 1;
  const int sy0 = ytab[Y0] & 0xFFFF;
  const int nrow = min((int)(ytab[ylast] & 0xFFFF) + 2 - sy0, lds_rows);
  const int n16 = sw16 >> 4, n4 = sw16 >> 2, rows = ylast - Y0 + 1;
  const int rowb = 4 * n4;                                 // staged row pitch in bytes
  if (tid < kResizeMaxRows) {       // rows past the band repeat its last row: the unrolled row loop reads valid entries
    const int y = Y0 + min(tid, rows - 1);
    const uint32_t yt = ytab[y];
    // bit 31: this row's upper source row is the previous output row's lower one (its horizontal sums are reused).
    // Only the low 24 bits of the weight reach v_mad_u32_u24, also through 256 - w.
    const uint32_t step1 = (tid > 0 && tid < rows && (yt & 0xFFFF) == (ytab[y - 1] & 0xFFFF) + 1u) ? 0x80000000u : 0u;
    ysh[2 * tid] = (uint32_t)(((int)(yt & 0xFFFF) - sy0) * rowb);     // LDS byte offset of the upper source row
    ysh[2 * tid + 1] = (yt >> 16) | step1;                             // weight of the lower source row
  }
  // a task = 4 output columns x 4 output rows; its table entries are requested one task ahead (the first one before
  // the band is staged), clamped instead of guarded so the loads are unconditional
  const int rgs = (rows + 3) >> 2, ntask = rgs * groups;
  int i = tid;
  int rg = magic_groups ? (int)__umulhi((uint32_t)min(i, ntask - 1), magic_groups) : min(i, ntask - 1);   // i / groups
  int gq = min(i, ntask - 1) - rg * groups;
  uint4 qs = xsel[gq], qw = xwxp[gq], qo = xoff[gq];
  for (int i = tid; i < nrow * n16; i += nthr) {
    const int r = (int)__umulhi((uint32_t)i, magic_n16), c = i - r * n16;   // i / n16 (exact for i < 2^16)
    const int gx = 16 * c, gy = sy0 + r;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (gy < sh && gx + 16 <= spitch) v = *reinterpret_cast<const uint4*>(s + (long long)gy * spitch + gx);
    reinterpret_cast<uint4*>(rt)[i] = v;
  }
  __syncthreads();
  for (; i < ntask; i += nthr) {
    const int x4 = 4 * gq, ry0 = 4 * rg;
    const uint32_t sel[4] = {qs.x, qs.y, qs.z, qs.w}, wxp[4] = {qw.x, qw.y, qw.z, qw.w};
    // pair offsets of the LOWER source row, relative to the upper row's LDS offset
    const uint32_t boff[4] = {qo.x + (uint32_t)rowb, (SHARED ? qo.x : qo.y) + (uint32_t)rowb,
                              (SHARED ? qo.x : qo.z) + (uint32_t)rowb, qo.w + (uint32_t)rowb};
    {
      const int in = min(i + nthr, ntask - 1);
      rg = magic_groups ? (int)__umulhi((uint32_t)in, magic_groups) : in;
      gq = in - rg * groups;
      qs = xsel[gq], qw = xwxp[gq], qo = xoff[gq];
    }
    const char* Tb = reinterpret_cast<const char*>(rt);
    uint8_t* dp = d + (long long)(Y0 + ry0) * L.pitch + x4;
    // the four rows' (LDS offset, weight) pairs in two 16-byte reads
    const uint4 ya = *reinterpret_cast<const uint4*>(ysh + 2 * ry0), yb = *reinterpret_cast<const uint4*>(ysh + 2 * ry0 + 4);
    const uint32_t yrb[4] = {ya.x, ya.z, yb.x, yb.z}, ywy[4] = {ya.y, ya.w, yb.y, yb.w};
    uint32_t hlow[4];                                      // horizontal sums of the previous row's LOWER source row
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const char* pr = Tb + yrb[j];
      // consecutive output rows usually step one source row: the lower row of the previous output row is this row's
      // upper row, its horizontal sums are reused (exactly the same integers)
      if (j == 0 || (int)ywy[j] >= 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t* pa = reinterpret_cast<const uint32_t*>(pr + boff[k] - rowb);      // row sy: p[cx], p[cx+1]
          // bytes (p0, p1) -> u16 pair, then w0*p0 + w1*p1 in one v_dot2_u32_u16 (<= 255 * 256: fits 16 bits)
          hlow[k] = udot2_u16(__builtin_amdgcn_perm(pa[1], pa[0], sel[k]), wxp[k], 0u);
        }
      }
      const uint32_t wy1 = ywy[j], wy0 = 256u - wy1;
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t* pc = reinterpret_cast<const uint32_t*>(pr + boff[k]);               // row sy+1
        const uint32_t h1 = udot2_u16(__builtin_amdgcn_perm(pc[1], pc[0], sel[k]), wxp[k], 0u);
        // weights sum to 256 * 256, so bits 16..23 hold the result (<= 255) without a clamp
        v[k] = mad_u24(hlow[k], wy0, mad_u24(h1, wy1, 32768u));
        hlow[k] = h1;
      }
      // byte 2 of each of the four sums -> one dword
      const uint32_t packed = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);
      if (ry0 + j < rows) *reinterpret_cast<uint32_t*>(dp) = packed;   // pitch % 16 == 0, pad bytes are never read as pixels
      dp += L.pitch;
    }
  }
}

// ------------------------------------------------------------------ K3+K4: FAST-9/16 score, NMS, border, candidate list
// Three dense phases per 128x32 tile instead of one divergent one:
//  1. prefilter, 4 px per lane in one dword (SWAR): any arc of 9 contains two ADJACENT cardinal ring pixels
//     ((0,3),(3,0),(0,-3),(-3,0)), so "some adjacent cardinal pair is all-brighter or all-darker" is necessary.
//     Per-byte threshold tests use v_lerp_u8 as a carry-free byte adder: lerp(p, ~c) = (p + 255 - c) >> 1, and a
//     second lerp against a constant puts the compare result in bit 7 of each byte.  Conservative (superset).
//  2. survivors (compacted in LDS, one list per polarity) get the exact cornerScore<16>: max over the 16 arcs of the
//     minimum over the arc, through min3/max3 window networks on the raw ring pixels; score > threshold <=> FAST_t's
//     9-contiguous test.  Scores go into a zeroed score tile.
//  3. strict 3x3 NMS + runByImageBorder, dense on 4 scores per lane (two passes over the tile's outputs).
// The kernel is VALU-issue bound: what counts is the number of vector instructions.
constexpr int HX = 16, HY = 4;                     // pixel-tile halo: rows start 16-byte aligned (x0 - 16)
constexpr int PW2 = TW + 2 * HX, PH2 = TH + 2 * HY;  // 96 x 40
constexpr int SCO = 4;                              // byte offset of the score tile inside its LDS array
// the score tile has the pixel tile's geometry (pitch PW2, same origin): one index serves both arrays
constexpr int GPR = TW / 4 + 2;                          // 4-px groups per score row: tile x = 4g-4 .. 4g-1
constexpr int kList1Cap = 2 * SW * SH;             // brighter-type survivors from the front, darker from the back
static_assert(kFastT % 2 == 0, "prefilter constants assume an even FAST threshold (cv::ORB default 20)");

__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) {   // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// cornerScore<16> restricted to one polarity (a pixel cannot have both a brighter and a darker arc of 9):
// max over the 16 arcs of the minimum of e over the arc, e = p - v (brighter) or v - p (darker).
template <bool BRIGHT, int PITCH = PW2>
__device__ __forceinline__ int fast_score_pol(const uint8_t* p, int tau) {
  constexpr int off[16] = {3 * PITCH + 0,  3 * PITCH + 1,  2 * PITCH + 2,  1 * PITCH + 3,  0 * PITCH + 3, -1 * PITCH + 3,
                           -2 * PITCH + 2, -3 * PITCH + 1, -3 * PITCH + 0, -3 * PITCH - 1, -2 * PITCH - 2, -1 * PITCH - 3,
                           0 * PITCH - 3,  1 * PITCH - 3,  2 * PITCH - 2,  3 * PITCH - 1};
  // brighter: max over arcs of min(p - v) = (max over arcs of min p) - v ; darker: max of min(v - p) = v - (min of max p):
  // the window networks run on the raw ring pixels and v is applied once at the end.
  int q[16];
#pragma unroll
  for (int k = 0; k < 16; k++) q[k] = (int)p[off[k]];
  int m3[16];
#pragma unroll
  for (int k = 0; k < 16; k++)
    m3[k] = BRIGHT ? min(min(q[k], q[(k + 1) & 15]), q[(k + 2) & 15]) : max(max(q[k], q[(k + 1) & 15]), q[(k + 2) & 15]);
  int W = BRIGHT ? -1 : 1000;
#pragma unroll
  for (int k = 0; k < 16; k++) {   // arc k .. k+8
    const int w9 = BRIGHT ? min(min(m3[k], m3[(k + 3) & 15]), m3[(k + 6) & 15])
                          : max(max(m3[k], m3[(k + 3) & 15]), m3[(k + 6) & 15]);
    W = BRIGHT ? max(W, w9) : min(W, w9);
  }
  const int v = p[0];
  const int A = BRIGHT ? W - v : v - W;
  return A > tau ? A - 1 : 0;   // = max(t, A, B) - 1 for corners (tau = t), 0 otherwise; tau > t keeps score >= tau only
}

// LDS operations of one wave execute in issue order: a wavefront-scope fence (compiler ordering + lgkmcnt wait) is all
// that wave-private LDS data needs between a write by one lane and a read by another
#define MSF_WAVE_SYNC()                                        \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)

// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts + row broadcasts, the gfx9 scan idiom)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// ordered, wave-aggregated reservation on a packed LDS counter (low 16 bits / high 16 bits = two lists):
// `mine` holds this lane's two counts packed the same way; returns the packed first slots of this lane.
__device__ __forceinline__ uint32_t reserve_packed(uint32_t mine, uint32_t* counter, int lane) {
  const uint32_t incl = wave_incl_scan(mine);
  const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
  uint32_t base = 0;
  if (total) {
    if (lane == 0) base = atomicAdd(counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
  }
  return base + incl - mine;
}

// 8 waves (two per SIMD) on a 128 x 32 tile: 11.8 ms per 2048 720p frames.  Measured alternatives: 64x32 / 256 threads
// 12.6, 64x64 / 512 12.0, 64x128 / 1024 13.5, 128x64 / 1024 12.9, 128x48 / 512 15.5, 192x32 / 512 16.0, 256x32 / 1024 13.2;
// 320- or 768-thread workgroups sit unevenly on the 4 SIMDs.  The wider tile halves the share of the two
// single-pixel edge groups per score row and of the halo columns.
constexpr int kFastThreads = 512;

struct FastSmem {
  __attribute__((aligned(16))) uint8_t px[PW2 * PH2];
  __attribute__((aligned(16))) uint8_t sc[PW2 * PH2 + 2 * SCO];   // one dword of slack either side
  uint16_t list1[kList1Cap];
  uint2 llist[kTileCandCap];
  uint32_t nbd, lcount, gbase;   // nbd: brighter count (low 16) | darker count (high 16)
};

// ---- where the candidate list of (frame, level) lives (r05): cmap[slot * 8 + level] = (first entry, capacity) in the
// candidate arrays -- the level's primary list inside the frame's work row, or a block of the pool (orb_pipeline.h)
__global__ __launch_bounds__(256) void k_cand_reset(OrbGeometry g, OrbPrimLists prim, uint2* cmap_rows, uint32_t* pool_cnt,
                                                    int n_frames, int dense) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *pool_cnt = 0u;
  if (i >= n_frames * kOrbLevels) return;
  const int row = i / kOrbLevels, l = i - row * kOrbLevels;
  if (l >= g.nlevels) { cmap_rows[i] = make_uint2(0u, 0u); return; }
  const OrbLevelInfo& L = g.lv[l];
  // a dense call (every level at fastThreshold): frame `row` gets a full-capacity region of the pool (host-checked: it fits)
  cmap_rows[i] = dense ? make_uint2((uint32_t)(g.pool_base + (long long)row * g.cand_total + L.cand_off), (uint32_t)L.cand_cap)
                       : make_uint2((uint32_t)((long long)row * g.prim_total + prim.off[l]), (uint32_t)prim.cap[l]);
}
// one lane: (slot, level) moves to a pool region of its full capacity (before its dense pass); false: the pool is exhausted
__device__ __forceinline__ bool cand_take_block(const OrbGeometry& g, uint2* cmap, uint32_t* pool_cnt, int idx, int l) {
  const uint32_t cap = (uint32_t)g.lv[l].cand_cap;              // (a multiple of 16: regions stay 16-byte aligned)
  const uint32_t at = atomicAdd(pool_cnt, cap);
  if (at > g.pool_entries || cap > g.pool_entries - at) return false;
  cmap[idx] = make_uint2((uint32_t)(g.pool_base + (long long)at), cap);
  return true;
}

// One tile of level l of frame fi with score threshold tau (even, >= kFastT): emits exactly the strict 3x3 maxima whose
// score is >= tau.  With tau = kFastT that is FAST_t<16> + NMS; with a larger tau it is the subset retainBest(2N) can
// still keep: a pixel whose score is below tau gets 0 in the score tile, which is what a maximum with score >= tau
// needs to know about it (it loses), and every pixel with score >= tau passes the prefilter at tau and is scored exactly.
// All threads of the workgroup must call this together; the LDS block may be reused after the call returns.
__device__ __forceinline__ void fast_tile(const OrbGeometry& g, const FrameSrc& src, const uint8_t* pyr, int fi, int l,
                                          int t, int tau, uint32_t* cand_cnt, uint32_t* cand_key, uint8_t* cand_sc,
                                          const uint2* cmap, FastSmem& S_) {
  uint8_t* px = S_.px;
  uint8_t* sc = S_.sc;
  uint16_t* list1 = S_.list1;
  uint2* llist = S_.llist;
  const int slot = src.slot0 + fi;
  const OrbLevelInfo L = g.lv[l];
  const int x0 = kTileX0 + (t % L.tiles_x) * TW, y0 = kTileY0 + (t / L.tiles_x) * TH;
  int pitch;
  const uint8_t* img = level_ptr(g, src, pyr, fi, l, &pitch);
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) { S_.nbd = 0; S_.lcount = 0; }
  const uint32_t lerp_bright = 0x01010101u * (uint32_t)(128 - tau / 2);          // L + K >= 256  <=>  L >= 128 + tau/2
  const uint32_t lerp_not_dark = 0x01010101u * (uint32_t)(255 - (254 - tau) / 2);

  // stage the pixel tile through aligned 16-byte loads (rows are 16-byte aligned: pitch % 16 == 0, x0 % 64 == 0);
  // zero the score tile
  for (int i = tid; i < (PW2 / 16) * PH2; i += kFastThreads) {
    const int r = i / (PW2 / 16), c = i % (PW2 / 16);
    const int gx = x0 - HX + 16 * c, gy = y0 - HY + r;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (gx >= 0 && gx + 16 <= pitch && gy >= 0 && gy < L.h)
      v = *reinterpret_cast<const uint4*>(img + (long long)gy * pitch + gx);
    reinterpret_cast<uint4*>(px)[i] = v;
  }
  for (int i = tid; i < (PW2 * PH2 + 2 * SCO) / 4; i += kFastThreads) reinterpret_cast<uint32_t*>(sc)[i] = 0;
  __syncthreads();

  // phase 1: cardinal prefilter on 4 px per lane
  const uint32_t* T = reinterpret_cast<const uint32_t*>(px);
  // Scored domain: FAST_t scores rows/cols 3 .. dim-4, but runByImageBorder(31) discards everything outside
  // [31, dim-31), so only those pixels and their NMS neighbours ([30, dim-30)) can matter; the rest keeps score 0.
  constexpr int kLo = kEdge - 1;
  const int txlo = max(-1, kLo - x0), txhi = min(TW, L.w - kLo - 1 - x0);   // scored tile-x range
  const bool inner1 = txlo == -1 && txhi == TW && y0 - 1 >= kLo && y0 + TH < L.h - kLo;
  for (int i0 = 0; i0 < GPR * SH; i0 += kFastThreads) {
    const int i = i0 + tid;
    if (i0 + (tid & ~63) >= GPR * SH) continue;       // wave-uniform: the last pass does not fill all waves
    const int ic = i < GPR * SH ? i : GPR * SH - 1;   // idle lanes recompute the last task, masked below
    const int sr = ic / GPR, gq = ic % GPR;        // score row, 4-px group
    const int ty = sr - 1, tx0 = 4 * gq - 4;
    const int b = ((ty + HY) * PW2 + tx0 + HX) >> 2;
    const uint32_t C = T[b], Lf = T[b - 1], Rt = T[b + 1];
    const uint32_t U = T[b - 3 * (PW2 / 4)], D = T[b + 3 * (PW2 / 4)];
    const uint32_t W3 = __builtin_amdgcn_alignbyte(C, Lf, 1);   // pixels x-3 .. x
    const uint32_t E3 = __builtin_amdgcn_alignbyte(Rt, C, 3);   // pixels x+3 .. x+6
    const uint32_t nC = ~C;
    const uint32_t l0 = __builtin_amdgcn_lerp(D, nC, 0), l4 = __builtin_amdgcn_lerp(E3, nC, 0);
    const uint32_t l8 = __builtin_amdgcn_lerp(U, nC, 0), l12 = __builtin_amdgcn_lerp(W3, nC, 0);
    const uint32_t b0 = __builtin_amdgcn_lerp(l0, lerp_bright, 0), b4 = __builtin_amdgcn_lerp(l4, lerp_bright, 0);
    const uint32_t b8 = __builtin_amdgcn_lerp(l8, lerp_bright, 0), b12 = __builtin_amdgcn_lerp(l12, lerp_bright, 0);
    const uint32_t n0 = __builtin_amdgcn_lerp(l0, lerp_not_dark, 0), n4 = __builtin_amdgcn_lerp(l4, lerp_not_dark, 0);
    const uint32_t n8 = __builtin_amdgcn_lerp(l8, lerp_not_dark, 0), n12 = __builtin_amdgcn_lerp(l12, lerp_not_dark, 0);
    // the score tile spans tile +- 1 (group 0 holds only x = -1, group 17 only x = TW)
    uint32_t vm = 0;
    if (inner1) {   // uniform: the whole score tile lies inside the scored domain
      vm = gq == 0 ? 0x80000000u : gq == GPR - 1 ? 0x00000080u : 0x80808080u;
      vm = i < GPR * SH ? vm : 0u;
    } else {
      const int gy = y0 + ty;
      int first = txlo - tx0, last = txhi - tx0;         // valid bytes: first .. last
      first = first < 0 ? 0 : first;
      last = last > 3 ? 3 : last;
      if (i < GPR * SH && gy >= kLo && gy < L.h - kLo && first <= last)
        vm = (0x80808080u << (8 * first)) & (0x80808080u >> (8 * (3 - last)));
    }
    const uint32_t cb = ((b0 | b8) & (b4 | b12)) & vm;
    const uint32_t cd = ~((n0 & n8) | (n4 & n12)) & vm;
    if (__ballot((cb | cd) != 0u) == 0ull) continue;   // flat region: nothing to append for this wave
    const uint32_t e0 = (uint32_t)b << 2;   // byte index of the group's first pixel in the pixel / score tile
    // lane order = x order, so phase 2's LDS reads stay bank-friendly
    const uint32_t slots = reserve_packed(__popc(cb) | (__popc(cd) << 16), &S_.nbd, lane);
    uint32_t k = slots & 0xFFFFu;
    if (cb & 0x80u) list1[k++] = (uint16_t)e0;
    if (cb & 0x8000u) list1[k++] = (uint16_t)(e0 + 1);
    if (cb & 0x800000u) list1[k++] = (uint16_t)(e0 + 2);
    if (cb & 0x80000000u) list1[k++] = (uint16_t)(e0 + 3);
    k = kList1Cap - 1 - (slots >> 16);
    if (cd & 0x80u) list1[k--] = (uint16_t)e0;
    if (cd & 0x8000u) list1[k--] = (uint16_t)(e0 + 1);
    if (cd & 0x800000u) list1[k--] = (uint16_t)(e0 + 2);
    if (cd & 0x80000000u) list1[k--] = (uint16_t)(e0 + 3);
  }
  __syncthreads();

  // phase 2: exact score of the survivors, one polarity per loop, written into the (zeroed) score tile
  const uint32_t mb = S_.nbd & 0xFFFFu, md = S_.nbd >> 16;
  if (mb | md) {   // uniform
    for (uint32_t i = tid; i < mb; i += kFastThreads) {
      const int e = list1[i];
      const int s = fast_score_pol<true>(&px[e], tau);
      if (s) sc[SCO + e] = (uint8_t)s;
    }
    for (uint32_t i = tid; i < md; i += kFastThreads) {
      const int e = list1[kList1Cap - 1 - i];
      const int s = fast_score_pol<false>(&px[e], tau);
      if (s) sc[SCO + e] = (uint8_t)s;
    }
    __syncthreads();

    // phase 3: strict 3x3 NMS + runByImageBorder(31), dense on 4 scores per lane: "c > n" per byte is bit 7 of
    // lerp(c, ~n) = (c + 255 - n) >> 1; non-corners hold 0 and can never be strictly greater.  The 128 x 32 outputs
    // are exactly 32 x 32 dwords = two passes of the workgroup.
    const uint32_t* S = reinterpret_cast<const uint32_t*>(sc) + SCO / 4;
    constexpr int SPD = PW2 / 4;                                    // score-tile pitch in dwords
    const int txlo3 = max(0, kEdge - x0), txhi3 = min(TW - 1, L.w - kEdge - 1 - x0);   // 31-px border
    const bool inner3 = txlo3 == 0 && txhi3 == TW - 1;
    for (int i = tid; i < (TW / 4) * TH; i += kFastThreads) {
      const int ty = i / (TW / 4), j = i % (TW / 4);
      const int w = (ty + HY) * SPD + HX / 4 + j;
      const uint32_t C = S[w];
      uint32_t keep = 0;
      const int gy = y0 + ty;
      const bool live = C != 0 && gy >= kEdge && gy < L.h - kEdge;
      if (__ballot(live) == 0ull) continue;              // no corner in these 256 score columns
      if (live) {
        const uint32_t Cl = S[w - 1], Cr = S[w + 1];
        const uint32_t U = S[w - SPD], Ul = S[w - SPD - 1], Ur = S[w - SPD + 1];
        const uint32_t D = S[w + SPD], Dl = S[w + SPD - 1], Dr = S[w + SPD + 1];
        keep = __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(C, Cl, 3), 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(Cr, C, 1), 0);
        keep &= __builtin_amdgcn_lerp(C, ~U, 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(U, Ul, 3), 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(Ur, U, 1), 0);
        keep &= __builtin_amdgcn_lerp(C, ~D, 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(D, Dl, 3), 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(Dr, D, 1), 0);
        if (!inner3) {   // uniform: only tiles that touch the 31-px border mask columns
          int first = txlo3 - 4 * j, last = txhi3 - 4 * j;
          first = first < 0 ? 0 : first;
          last = last > 3 ? 3 : last;
          keep &= first <= last ? (0x80808080u << (8 * first)) & (0x80808080u >> (8 * (3 - last))) : 0u;
        } else {
          keep &= 0x80808080u;
        }
      }
      const uint32_t mine = __popc(keep);
      uint32_t k = reserve_packed(mine, &S_.lcount, lane);
      const uint32_t key = ((uint32_t)gy << 16) | (uint32_t)(x0 + 4 * j);
      if (keep & 0x80u) llist[k++] = make_uint2(key, C & 0xFFu);
      if (keep & 0x8000u) llist[k++] = make_uint2(key + 1, (C >> 8) & 0xFFu);
      if (keep & 0x800000u) llist[k++] = make_uint2(key + 2, (C >> 16) & 0xFFu);
      if (keep & 0x80000000u) llist[k++] = make_uint2(key + 3, C >> 24);
    }
    __syncthreads();
    const uint32_t n = S_.lcount;
    if (n != 0) {   // uniform
      if (tid == 0) S_.gbase = atomicAdd(&cand_cnt[slot * kOrbLevels + l], n);
      __syncthreads();
      const uint32_t base = S_.gbase;
      // structure of arrays: the retainBest threshold pass reads every score (1 byte) but only a few hundred keys
      const uint2 cm = cmap[slot * kOrbLevels + l];
      uint32_t* outk = cand_key + cm.x;
      uint8_t* outs = cand_sc + cm.x;
      for (uint32_t i = tid; i < n; i += kFastThreads)
        if (base + i < cm.y) {
          const uint2 e = llist[i];
          outk[base + i] = e.x;
          outs[base + i] = (uint8_t)e.y;
        }
    }
  }
}

// K3+K4 over every tile of every level of every frame of the batch at fastThreshold (MSF_FLAG_FAST_DENSE).
__global__ __launch_bounds__(kFastThreads) void k_fast(OrbGeometry g, FrameSrc src, const uint8_t* pyr,
                                                       uint32_t* cand_cnt, uint32_t* cand_key, uint8_t* cand_sc,
                                                       const uint2* cmap) {
  __shared__ FastSmem sm;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so workgroup b
  // takes tile start(b % 8) + b / 8 of the flattened (frame, tile) list: one XCD walks one contiguous run of tiles and
  // the cache lines neighbouring tiles share are fetched into one L2 instead of eight.  Speed only, any placement works.
  int G;
  {
    const uint32_t total = gridDim.x, lin = blockIdx.x;
    const uint32_t xcd = lin & 7u, q = total >> 3, r = total & 7u;
    G = (int)(xcd * q + (xcd < r ? xcd : r) + (lin >> 3));
  }
  const int fi = G / g.total_tiles, bt = G - fi * g.total_tiles;
  int l = 0;
#pragma unroll
  for (int i = 1; i < kOrbLevels; i++)
    if (i < g.nlevels && bt >= g.lv[i].tile_base) l = i;
  fast_tile(g, src, pyr, fi, l, bt - g.lv[l].tile_base, kFastT, cand_cnt, cand_key, cand_sc, cmap, sm);
}

// ------------------------------------------------------------------ K2+K3+K4, streaming form: one wave walks a column strip
// The tile kernels above are bound by workgroup dispatch and by their barriers once the scoring work is gone (measured:
// 1.9 ms to dispatch the 10^6 tiles of a 1024-pair batch, 6 ms of phase latency).  Here ONE WAVE walks a column strip
// of a level, a 256-px window (lane i holds 4 px), top to bottom, with no barrier:
//  * every pixel is fetched once, as one dword per lane and row, four rows ahead of its use (register queue q0..q3);
//  * the rows live in a wave-private LDS ring of 16 rows;
//  * the cardinal prefilter of row y compares it with row y+3 and with its own pixels three to the left / right (DPP); its
//    comparison with row y-3 is the one row y-3 made three rows earlier (see STREAM_PRE);
//  * groups with a survivor leave one record; every few rows (or when the record list fills) the records are expanded
//    into one pixel list, scored exactly (darker-type pixels on the complemented values, so one routine serves both
//    polarities), the scores go into a second ring and, as (address, row) hits, into a list; strict 3x3 NMS +
//    runByImageBorder run over the hits whose three score rows are final; kept corners are buffered and appended to
//    the level's candidate list with one global atomic per buffer flush.
// Emits exactly the maxima with score >= tau of the strip's pixels, like fast_tile(tau).
//
// RESIZE (r03): the strips cover the whole level and the walker of level l - 1 also MAKES level l (K2,
// cv::resize INTER_LINEAR_EXACT as in k_resize, the same host tables): the source rows it needs are the rows in its
// ring.  A lane owns one group of 4 output columns whose taps lie inside the strip's window; when source row y arrives
// the lane forms its four horizontal 8.8 sums from the ring (two 8-byte LDS reads, v_perm, v_dot2), and if an output
// row Y has rows (y - 1, y) as its taps it blends them with the previous row's sums and stores one dword.  Every pixel
// of level l - 1 is then read from HBM once per step (plus the strips' halos) instead of once by k_resize and once by
// the FAST pass: 17.2 -> 10.5 GB per 1024-pair 720p step for pyramid + FAST.
//
// Threshold refinement inside the launch.  The strips of a level whose index is a multiple of 4 -- a quarter of the level,
// spread over it -- come first in the launch order and run at the level's first threshold (tau_unit); each adds the scores
// of the corners it emitted to a per-(frame, level) histogram and counts itself done.  A strip of the other three quarters
// waits for that count, raises tau to the largest multiple of 4 at which the quarter still holds margin % of its share of
// 2N and publishes it with an atomic max.  Whatever threshold a strip ran at, it finds every maximum >= its own tau, so
// every maximum >= the final tau[idx] (the maximum over the strips) is in the list; k_fast_check counts those and a level
// that falls short is redone densely: the result is the dense one bit for bit whatever the thresholds were.
constexpr int kTauBins = 64;             // score histogram bins of width 4 (the threshold sample, the emitted corners)
// State of a (slot, level) while the walker launch runs: kQStat words, zeroed (hipMemsetAsync) before the launch, every
// access an agent-scope atomic.  It is both the statistics the thresholds come from and the dependency tracking of the
// one-launch form (see k_walk):
//   [0, 64)   histogram of the FAST scores of the corners the level's sampled quarter emitted
//   kQDone    strips of the quarter that are done (their histogram adds included)
//   kQTau     the threshold in force: the largest refined threshold any strip published (atomicMax; 0 = none yet)
//   kQFirst   kQReady | the first estimate, stored by the (frame, level)'s threshold unit once the level exists
//   kQAll     strips of the level that are done: at wk_nx * wk_ny every pixel of level l + 1 has been written
constexpr int kQDone = kTauBins, kQTau = kTauBins + 1, kQFirst = kTauBins + 2, kQAll = kTauBins + 3;
constexpr int kQStat = kTauBins + 4;
constexpr uint32_t kQReady = 0x80000000u;
constexpr int RK = 16;                   // ring rows (power of two)
constexpr int kWkMaxPx = 244;            // strip pitch <= 244: a group of 4 output px reads two aligned 8-byte pairs <= 8 B apart
constexpr int kWkMaxRows = 240;          // owned rows per strip at most (a strip's emit table lives in four registers per lane)
// Owned rows per strip aimed at.  With the levels as separate launches 80 rows measured best (r03: 48 / 64 / 80 / 96 / 112
// rows 8.19 / 7.97 / 7.83 / 7.88 / 7.96 ms per 720p step: taller strips lengthen every launch's tail).  In the one-launch
// form there is one tail, and what counts is the work a strip repeats -- 8 halo rows, its prologue, its drain:
// 64 / 80 / 112 / 144 / 180 / 240 rows measured 5.47 / 5.24 / 4.97 / 4.86 / 4.78 / 4.79 ms for the stage at 1280 x 720
// (1024 pairs) and 9.92 (80) / 9.45 (112) / 8.80 (144) / 8.62 (240) ms at 640 x 480 (4096 pairs).  Handles for small
// batches keep short strips: their calls are latency-bound chains of (quarter, rest) x 8 levels.
constexpr int kWkRowsTall = 240, kWkRowsShort = 80, kWkTallMinSlots = 4 * 64;
constexpr int kTauSites = 1024;          // sampled runs of 4 px per (frame, level) of the threshold sampler (512 / 2048 / 4096: slower)
constexpr int kSGCap = 288, kSPCap = 256, kSHCap = 64, kSOCap = 64;   // 10 240 B of LDS per wave: exactly 16 waves per CU
constexpr uint32_t kSGFlush = 32;        // records that trigger a flush at the end of a four-step group (<= 4 x 64 more arrive)
static_assert(kSGCap >= (int)kSGFlush + 256, "record list capacity");
// A row put at step s overwrites rel row s - 13; pending records read pixel rows >= last_flush - 2 and pending NMS
// score rows >= last_flush - 1: the flush interval (a multiple of the four-step group) must stay below 11 rows.
constexpr int kFlushRows = 8;
static_assert(kFlushRows < RK - 5 && kFlushRows % 4 == 0, "ring too short for the flush interval");
// kTau2MinStrips: sampled strips a level needs for its refinement / for the prediction of the level below.  1 since the
// tall strips (a 640 x 480 frame has levels of two to four strips): 640 x 480 stage 8.93 -> 8.45 ms, 1280 x 720 unchanged,
// no additional dense pass in either.
constexpr int kTau2MarginPct = 200, kTau2MinStrips = 1;
constexpr int kTau2MarginPctMany = 150, kTau2ManyStrips = 8;

struct StreamSmem {
  __attribute__((aligned(16))) uint8_t px[RK * 256];
  __attribute__((aligned(16))) uint8_t sc[RK * 256];
  uint32_t g[kSGCap];        // per byte j: bit 7 = px j passed the brighter-type prefilter, bit 6 = the darker-type one; in
                             // the low six bits of bytes 0 / 1 / 2: lane, rel row & 63, rel row >> 6
  uint16_t p[kSPCap];        // pixel entries: byte in row | ring row << 8 | darker-type << 12
  uint16_t h[kSHCap];        // scored corners: byte in row | ring row << 8 (the rel row follows from the ring row)
  uint32_t opk[kSOCap];      // kept corners waiting for the next flush to global memory: byte in row | rel row << 8 | score << 16
};
static_assert(sizeof(StreamSmem) == 10240, "16 waves per CU need 10 240 B per wave");

// cornerScore<16> of the pixel at byte x of ring row r0 (rows wrap modulo RK), both polarities in one routine:
// darker-type = brighter-type on 255 - p.  Returns score (>= tau) or 0.
__device__ __forceinline__ int stream_score(const uint8_t* ring, uint32_t r0, uint32_t x, bool dark, int tau) {
  uint32_t rb[7];
#pragma unroll
  for (int d = 0; d < 7; d++) rb[d] = (((r0 + (uint32_t)(d + RK - 3)) & (uint32_t)(RK - 1)) << 8) + x;
  // ring offsets (dx, dy), k = 0..15: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
  constexpr int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const uint32_t flip = dark ? 255u : 0u;
  int q[16];
#pragma unroll
  for (int k = 0; k < 16; k++) q[k] = (int)((uint32_t)ring[rb[dy[k] + 3] + dx[k]] ^ flip);
  const int v = (int)((uint32_t)ring[rb[3]] ^ flip);
  int m3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) m3[k] = min(min(q[k], q[(k + 1) & 15]), q[(k + 2) & 15]);
  int W = -1;
#pragma unroll
  for (int k = 0; k < 16; k++) W = max(W, min(min(m3[k], m3[(k + 3) & 15]), m3[(k + 6) & 15]));
  const int A = W - v;
  return A > tau ? A - 1 : 0;
}

// ---- the one-launch form: units, their order, and what a unit may wait for
// A walker launch covers levels [l_lo, l_hi] of n_frames frames.  Per (frame, level) it holds one THRESHOLD unit (the
// first estimate of the level's FAST threshold: predicted from the level above or sampled, tau_unit) and one unit per
// strip; a unit is a workgroup of one wave.  Workgroups b, b + 8, ... share an XCD (observed placement, speed only): XCD
// x takes the frames [x n / 8, (x + 1) n / 8) and walks, level by level, first the threshold units of its frames, then
// the sampled quarter of all their strips (strip index a multiple of 4), then the rest -- so the lines neighbouring
// strips share stay in one L2, and whatever a unit needs was started long before it.
// Dependencies (chain = 1: the walker of level l - 1 makes level l):
//   threshold unit (f, l), l > 0 : all strips of (f, l - 1) done            (kQAll of level l - 1: level l exists)
//   strip of (f, l)              : the threshold of (f, l) published         (kQFirst; implies the line above)
//   non-quarter strip, dyn       : the quarter of (f, l) done                (kQDone: the refined threshold)
// Every wait is for units with a LOWER flat workgroup index, so with workgroups started in index order the lowest
// unfinished one never waits and the launch drains; the waits are bounded all the same (kSpinMax polls of ~2 us): a unit
// that gives up flags its frame (status, surfaced as n_out = -1), raises the launch's abort word -- every other waiter
// then leaves at its next poll -- and the grid always terminates.  Results never depend on placement or timing: pixels
// are handed over with write-through stores, a drained store queue, an agent-scope counter, and an agent-scope acquire
// on the reading side; thresholds may differ from run to run in principle, the candidate lists they lead to contain
// every corner retainBest(2N) can keep either way (k_fast_check).
constexpr uint32_t kSpinMax = 1u << 19;


__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a unit gives up: its frame has no valid result, every waiter of the launch leaves
__device__ __forceinline__ void unit_stall(uint32_t* status, int slot, uint32_t* abort_word, int lane) {
  if (lane == 0) {
    atomicOr(&status[slot], kStatusOverflow);
    __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // word 1 is STICKY: counted up, never zeroed between calls.  The host reads it behind an asynchronous copy, possibly
    // many calls later (a caller may enqueue dozens of calls without a host wait: word 0 is long zero again by then).
    atomicAdd(abort_word + 1, 1u);
  }
}

// one strip: strip ts of level l of frame fi
template <bool RESIZE>
__device__ __forceinline__ void walk_strip(StreamSmem& sm, const OrbGeometry& g, const FrameSrc& src, uint8_t* pyr,
                                           const uint32_t* __restrict__ tab, uint32_t* qstat, uint32_t* cand_cnt,
                                           uint32_t* cand_key, uint8_t* cand_sc, const uint2* cmap, uint32_t* status,
                                           uint32_t* abort_word, int margin_pct, int dyn, int chain, const int l, const int fi,
                                           const int ts) {
  const int lane = threadIdx.x;
  const bool quarter = (ts & 3) == 0;
  const OrbLevelInfo L = g.lv[l];
  const int slot = src.slot0 + fi, idx = slot * kOrbLevels + l;
  const int sy = ts / L.wk_nx, sx = ts - sy * L.wk_nx;
  uint32_t* const qs = qstat + (size_t)idx * kQStat;
  const int n_strips = L.wk_nx * L.wk_ny;
  const int qa = (n_strips + 3) >> 2;            // strips of this level in the sampled quarter
  const bool count_me = dyn && quarter;          // single exit below: a quarter strip always reports itself done

  // ---- geometry of the strip
  const int xs = sx * L.wk_px;                    // x of lane 0's first px (multiple of 4)
  const int xb = xs + 4 * lane;
  const int R0 = sy * L.wk_rows, R1 = min(R0 + L.wk_rows, L.h);      // owned rows
  // FAST outputs: owned rows inside the 31-px border; a FAST-only walk covers just those
  const int ya = RESIZE ? R0 : max(R0, kEdge), yb = RESIZE ? R1 : min(R1, L.h - kEdge);
  const int ox0 = sx == 0 ? 0 : xs + 4, ox1 = sx == L.wk_nx - 1 ? L.w : xs + 4 + L.wk_px;   // owned columns
  const int y0 = ya - 1;                          // rel row r = y - y0; rel row 0 is the NMS neighbour of row ya
  int pitch;
  const uint8_t* img = level_ptr(g, src, pyr, fi, l, &pitch);
  // lanes right of the row's end re-read its last dword (weight-0 taps, never scored); rows outside the image re-read
  // its first / last row (warm-up above row 0, the idle tail steps, the weight-0 lower tap of the last output row)
  // Lanes whose pixels nobody needs -- the windows of neighbouring strips overlap by 256 - wk_px columns, of which FAST
  // needs 4 and the resize at most 16 -- re-read the nearest needed dword instead of their own: the overlap is then
  // fetched once (the neighbour runs at another time: what it shares with this strip has left the L2 by then).
  const int need_lo = RESIZE ? xs : max(ox0 - 4, xs);
  const int need_hi = sx == L.wk_nx - 1 ? pitch : (RESIZE ? xs + L.wk_px + 16 : ox1 + 4);     // exclusive, multiples of 4
  const int xl = min(max(xb, need_lo & ~3), ((need_hi + 3) & ~3) - 4);
  // wave-uniform row base (scalar registers) + the lane's 32-bit byte offset: the load takes its address as
  // s[base] + v_off, no vector arithmetic per row
  const uint32_t xoff = (uint32_t)(xl + 4 <= pitch ? xl : pitch - 4);
  const int ylim = L.h - 1;
  const uint32_t pitch_u = uniform_u32((uint32_t)pitch);       // a level is far below 4 GB: 32-bit row offsets
  const __amdgpu_buffer_rsrc_t img_rs = uniform_rsrc(img, (uint32_t)L.h * (uint32_t)pitch);
#define LOAD_ROW(y_) ((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(img_rs, xoff, (uint32_t)max(min((y_), ylim), 0) * pitch_u, 0))
  // Level 0 is the caller's frame: its first ten pixel rows are requested before anything else, together with the state
  // of the (frame, level) -- one memory latency for both.  A level the walker made is only read once that state says it
  // is complete (the threshold unit of this (frame, level) waited for it) and behind an acquire.
  const bool early = !(chain && l > 0);          // uniform
  uint32_t w0_ = 0, w1_ = 0, w2_ = 0, w3_ = 0, w4_ = 0, w5_ = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  if (early) {
    w0_ = LOAD_ROW(y0 - 3); w1_ = LOAD_ROW(y0 - 2); w2_ = LOAD_ROW(y0 - 1); w3_ = LOAD_ROW(y0); w4_ = LOAD_ROW(y0 + 1);
    w5_ = LOAD_ROW(y0 + 2);
    q0 = LOAD_ROW(y0 + 3); q1 = LOAD_ROW(y0 + 4); q2 = LOAD_ROW(y0 + 5); q3 = LOAD_ROW(y0 + 6);
  }

  // ---- threshold: the first estimate, raised from the quarter's exact corners.  The whole state of the (frame, level)
  // -- 64 bins and the four words behind them -- comes in with two independent loads (one latency).
  const bool need_q = dyn && !quarter && qa >= kTau2MinStrips;
  uint32_t qv = ld_agent(&qs[lane]);
  uint32_t qx = ld_agent(&qs[kTauBins + (lane & 3)]);
  {
    bool ready = false;
    for (uint32_t it = 0; it < kSpinMax; it++) {
      const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)qx, kQFirst - kTauBins);
      const uint32_t done = (uint32_t)__builtin_amdgcn_readlane((int)qx, kQDone - kTauBins);
      ready = (first & kQReady) != 0u && (!need_q || done >= (uint32_t)qa);
      if (ready || ld_agent(abort_word) != 0u) break;     // uniform
      __builtin_amdgcn_s_sleep(20);
      qv = ld_agent(&qs[lane]);
      qx = ld_agent(&qs[kTauBins + (lane & 3)]);
    }
    if (!ready) {
      unit_stall(status, slot, abort_word, lane);
      return;
    }
  }
  if (!early) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    w0_ = LOAD_ROW(y0 - 3); w1_ = LOAD_ROW(y0 - 2); w2_ = LOAD_ROW(y0 - 1); w3_ = LOAD_ROW(y0); w4_ = LOAD_ROW(y0 + 1);
    w5_ = LOAD_ROW(y0 + 2);
    q0 = LOAD_ROW(y0 + 3); q1 = LOAD_ROW(y0 + 4); q2 = LOAD_ROW(y0 + 5); q3 = LOAD_ROW(y0 + 6);
  }
  int tv = (int)((uint32_t)__builtin_amdgcn_readlane((int)qx, kQFirst - kTauBins) & ~kQReady);
  tv = max(tv, (int)__builtin_amdgcn_readlane((int)qx, kQTau - kTauBins));
  if (need_q && tv > kFastT) {
    uint32_t c = qv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {   // suffix sums: c = corners of the quarter with score >= 4 * lane
      const uint32_t up = __shfl_down(c, o);
      if (lane + o < 64) c += up;
    }
    // a level with many sampled strips gives a steadier estimate and takes the smaller margin; an explicit
    // MSF_ORB_TAU2_MARGIN_PCT applies to every level
    const uint32_t mp = (qa >= kTau2ManyStrips && margin_pct == kTau2MarginPct) ? (uint32_t)kTau2MarginPctMany : (uint32_t)margin_pct;
    const uint32_t den = 100u * (uint32_t)n_strips;
    const uint32_t need = (mp * 2u * (uint32_t)L.quota * (uint32_t)qa + den - 1u) / den;
    const unsigned long long ok = __ballot(c >= need && 4 * lane >= tv);
    if (ok) {
      const int t2 = 4 * (63 - __builtin_clzll(ok));
      if (t2 > tv) {
        tv = t2;
        if (lane == 0) atomicMax(&qs[kQTau], (uint32_t)t2);
      }
    }
  }
  const bool do_fast = tv > kFastT && L.tiles_x > 0;   // tau = fastThreshold: the level goes to the dense tile kernel
  const bool empty = !RESIZE && (!do_fast || ya >= yb || max(ox0, kEdge) >= min(ox1, L.w - kEdge));
  if (!empty) {
  const int r_last = yb - y0;                     // last rel row that can be scored (NMS neighbour of row yb - 1)
  // rows that are scored lie in [30, h - 30), rows that are output in [31, h - 31)
  const int s_lo = max(0, kEdge - 1 - y0), s_hi = min(r_last, L.h - kEdge - y0);
  const int o_lo = max(1, kEdge - y0), o_hi = min(r_last, L.h - kEdge - y0);      // output rel rows [o_lo, o_hi)
  uint32_t vm = 0, om = 0;
  if (do_fast) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = xb + j;
      // scored: owned columns and their NMS neighbours, with the 3-px ring inside the window
      if (x >= max(ox0 - 1, kEdge - 1) && x < min(ox1 + 1, L.w - (kEdge - 1)) && x >= xs + 3 && x < xs + 253) vm |= 0x80u << (8 * j);
      if (x >= max(ox0, kEdge) && x < min(ox1, L.w - kEdge)) om |= 0x80u << (8 * j);
    }
  }
  const int ox_lo = max(ox0, kEdge), ox_hi = min(ox1, L.w - kEdge);
  const uint32_t lerp_bright = 0x01010101u * (uint32_t)(128 - tv / 2);
  const uint32_t lerp_not_dark = 0x01010101u * (uint32_t)(255 - (254 - tv) / 2);
  uint32_t* const out_cnt = cand_cnt + idx;
  // where this (frame, level)'s list lives (read behind the wait for the level's threshold)
  const uint2 cm = cmap[idx];
  uint32_t* const outk = cand_key + cm.x;
  uint8_t* const outs = cand_sc + cm.x;
  const uint32_t out_cap = min(cm.y, (uint32_t)L.cand_cap);   // (a list never has more than the level's full capacity)
  uint32_t* pxw = reinterpret_cast<uint32_t*>(sm.px);
  uint32_t* scw = reinterpret_cast<uint32_t*>(sm.sc);
  uint8_t* pxb = sm.px;
  uint8_t* scb = sm.sc;

  // ---- resize state: this lane's group of 4 output columns of level l + 1
  uint32_t rsel[4] = {0, 0, 0, 0}, rwxp[4] = {0, 0, 0, 0}, roff0 = 0, roff3 = 0, em0 = 0, em1 = 0, em2 = 0, em3 = 0;
  __amdgpu_buffer_rsrc_t dst_rs = img_rs;         // level l + 1 of this frame (set below)
  uint32_t dxoff = 0;                             // the lane's byte offset inside an output row
  bool rz_lane = false;
  uint32_t dpitch = 0;
  if (RESIZE) {
    const OrbLevelInfo Ld = g.lv[l + 1];
    const int groups = (Ld.w + 3) >> 2;
    const uint4* xsel = reinterpret_cast<const uint4*>(tab + Ld.tab_off);
    const uint4* xwxp = xsel + groups;
    const uint4* xoff = xwxp + groups;
    const uint32_t* yemit = tab + Ld.tab_yemit;
    // straight from the kernel arguments (indexing the local copy Ld would put it in scratch memory)
    const int g_first = g.lv[l + 1].wk_xg[sx], g_end = g.lv[l + 1].wk_xg[sx + 1];   // (host-checked: wk_nx <= kWkMaxNx if fused)
    const int gq = min(g_first + lane, groups - 1);
    rz_lane = g_first + lane < g_end;
    const uint4 qsv = xsel[gq], qwv = xwxp[gq], qov = xoff[gq];
    rsel[0] = qsv.x; rsel[1] = qsv.y; rsel[2] = qsv.z; rsel[3] = qsv.w;
    rwxp[0] = qwv.x; rwxp[1] = qwv.y; rwxp[2] = qwv.z; rwxp[3] = qwv.w;
    // pair offsets relative to the window (host-checked: 0 <= off <= 248 for the groups of this strip)
    roff0 = rz_lane ? qov.x - (uint32_t)xs : 0u;
    roff3 = rz_lane ? qov.w - (uint32_t)xs : 0u;
    dpitch = uniform_u32((uint32_t)Ld.pitch);
    dst_rs = uniform_rsrc(pyr + (long long)slot * g.pyr_bytes + Ld.pix_off, (uint32_t)Ld.h * (uint32_t)Ld.pitch);
    dxoff = 4u * (uint32_t)gq;
    // emit entries of the source rows R0 - 4 + j, j = 64 k + lane in em<k>: output row | w1 << 16 | 1 << 31
    // if an output row has source rows (y, y + 1) as its taps and y is owned by this strip
    const int ra = R0 - 4 + lane, rb = ra + 64;
    if (ra >= R0 && ra < R1) em0 = yemit[ra];
    if (rb >= R0 && rb < R1) em1 = yemit[rb];
    if (rb + 64 >= R0 && rb + 64 < R1) em2 = yemit[rb + 64];
    if (rb + 128 >= R0 && rb + 128 < R1) em3 = yemit[rb + 128];
  }

  // the two 64-entry blocks of the emit table the current rows use (entries 64 em_k .. 64 em_k + 127)
  uint32_t em_a = em0, em_b = em1;
  int em_k = 0;
  // wave-uniform state
  uint32_t nG = 0, nH = 0, nO = 0;
  int nms_lo = 1, last_flush = -1;               // first rel row whose NMS is pending; rel row of the last flush

  // pixel row with ring index k (= rel row + 3) -> ring row k & 15; its score row is zeroed
#define PUT_ROW(k_, v_)                                                                      \
  do {                                                                                       \
    const int sl_ = (k_) & (RK - 1);                                                         \
    pxw[sl_ * 64 + lane] = (v_);                                                             \
    scw[sl_ * 64 + lane] = 0u;                                                               \
  } while (0)
  // the lane's four horizontal 8.8 sums of the pixel row with ring index k: w0 * p[cx] + w1 * p[cx + 1] per output column
#define RZ_HSUM(k_, h_)                                                                                                \
  do {                                                                                                                 \
    const uint8_t* rr_ = pxb + (((k_) & (RK - 1)) << 8);                                                               \
    const uint32_t* pa_ = reinterpret_cast<const uint32_t*>(rr_ + roff0);                                              \
    const uint32_t* pb_ = reinterpret_cast<const uint32_t*>(rr_ + roff3);                                              \
    const uint32_t a0_ = pa_[0], a1_ = pa_[1], b0_ = pb_[0], b1_ = pb_[1];                                             \
    h_[0] = udot2_u16(__builtin_amdgcn_perm(a1_, a0_, rsel[0]), rwxp[0], 0u);                                          \
    h_[1] = udot2_u16(__builtin_amdgcn_perm(a1_, a0_, rsel[1]), rwxp[1], 0u);                                          \
    h_[2] = udot2_u16(__builtin_amdgcn_perm(a1_, a0_, rsel[2]), rwxp[2], 0u);                                          \
    h_[3] = udot2_u16(__builtin_amdgcn_perm(b1_, b0_, rsel[3]), rwxp[3], 0u);                                          \
  } while (0)
  // j_ = index of the UPPER source row in the strip's emit table (row R0 - 4 + j_); hu_ / hl_ = sums of the upper / lower row
#define RZ_EMIT(j_, hu_, hl_)                                                                                          \
  do {                                                                                                                 \
    const int jj_ = (j_);                                                                                              \
    /* one select + one v_readlane (a ternary of two readlanes compiles to branches) */                                 \
    const uint32_t em_ = (uint32_t)__builtin_amdgcn_readlane((int)((jj_ >> 6) == em_k ? em_a : em_b), jj_ & 63);        \
    if ((int)em_ < 0) {                                                                                                \
      const uint32_t wy1_ = (em_ >> 16) & 0x1FFu, wy0_ = 256u - wy1_, rnd_ = 32768u;                                   \
      const uint32_t v0_ = mad_u24_s(hu_[0], wy0_, mad_u24_s(hl_[0], wy1_, rnd_));                                     \
      const uint32_t v1_ = mad_u24_s(hu_[1], wy0_, mad_u24_s(hl_[1], wy1_, rnd_));                                     \
      const uint32_t v2_ = mad_u24_s(hu_[2], wy0_, mad_u24_s(hl_[2], wy1_, rnd_));                                     \
      const uint32_t v3_ = mad_u24_s(hu_[3], wy0_, mad_u24_s(hl_[3], wy1_, rnd_));                                     \
      const uint32_t pk_ = __builtin_amdgcn_perm(v1_, v0_, 0x0c0c0602u) | __builtin_amdgcn_perm(v3_, v2_, 0x06020c0cu); \
      /* aux 16 = sc1: write-through, so that the level is in memory when this strip counts itself done (k_walk) */     \
      if (rz_lane) __builtin_amdgcn_raw_buffer_store_b32(pk_, dst_rs, dxoff, (em_ & 0xFFFFu) * dpitch, 16);             \
    }                                                                                                                  \
  } while (0)

  auto flush_out = [&]() {
    if (nO == 0) return;
    MSF_WAVE_SYNC();
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(out_cnt, nO);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < nO; i += 64) {
      const uint32_t pk = sm.opk[i], sc_ = pk >> 16;
      if (base + i < out_cap) {
        outk[base + i] = ((uint32_t)(y0 + (int)((pk >> 8) & 255u)) << 16) | (uint32_t)(xs + (int)(pk & 255u));
        outs[base + i] = (uint8_t)sc_;
      }
      if (count_me) atomicAdd(&qs[sc_ >> 2], 1u);     // the quarter's exact scores: what the other strips refine tau from
    }
    MSF_WAVE_SYNC();
    nO = 0;
  };

  // append (keep ? one output : nothing) of every lane, in lane order; the two half-waves one after the other, so
  // that a call adds at most 32 entries and the buffer (kSOCap >= 32) can always take them after a flush.
  // xl = byte of the corner inside the ring row, rr = its rel row.
  static_assert(kSOCap >= 32, "output buffer smaller than one half-wave");
  auto emit = [&](bool keep, uint32_t xl, uint32_t rr, uint32_t score) {
    if (__ballot(keep) == 0ull) return;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const bool kh = keep && (lane >> 5) == half;
      const unsigned long long bal = __ballot(kh);
      if (bal == 0ull) continue;
      const uint32_t cnt = (uint32_t)__popcll(bal);
      if (nO + cnt > (uint32_t)kSOCap) flush_out();
      if (kh) sm.opk[nO + mbcnt64(bal)] = xl | (rr << 8) | (score << 16);
      MSF_WAVE_SYNC();
      nO += cnt;
    }
  };

  // scores everything recorded so far (rel rows <= s), then NMS of rel rows [nms_lo, s - 1]
  auto flush = [&](int s) {
    // the hit list was rebuilt from one row's scores at the end of the last flush: if that row held more corners than
    // the list, this NMS is the dense one
    bool overflow = nH > (uint32_t)kSHCap;
    // The records are expanded into the pixel-entry list 64 at a time and the list is scored in full rounds of 64 whenever
    // it could not take the next round's entries (32 records at a time went through expansion + scoring with the lanes
    // 40 % used).  A record holds up to 8 entries (a pixel can pass both polarity prefilters): a round whose 64 records
    // hold more than the list takes 32 of them.
    uint32_t nP = 0;
    auto score_pending = [&]() {
      MSF_WAVE_SYNC();
      for (uint32_t i0 = 0; i0 < nP; i0 += 64) {
        const uint32_t i = i0 + lane;
        int sv = 0;
        uint32_t pe = 0;
        if (i < nP) {
          pe = sm.p[i];
          sv = stream_score(pxb, (pe >> 8) & 15u, pe & 255u, (pe & 0x1000u) != 0u, tv);
          if (sv) scb[pe & 0xFFFu] = (uint8_t)sv;
        }
        const unsigned long long bal = __ballot(sv != 0);
        if (bal) {
          const uint32_t cnt = (uint32_t)__popcll(bal);
          if (overflow || nH + cnt > (uint32_t)kSHCap) overflow = true;
          else {
            if (sv) sm.h[nH + mbcnt64(bal)] = (uint16_t)(pe & 0x0FFFu);
            nH += cnt;
          }
        }
      }
      MSF_WAVE_SYNC();
      nP = 0;
    };
    for (uint32_t c0 = 0; c0 < nG;) {
      uint32_t e0 = 0, mb = 0, md = 0;
      if (c0 + lane < nG) {
        const uint32_t rec = sm.g[c0 + lane];
        // only the low four bits of the rel row matter here (ring row = (rel row + 3) mod 16)
        // pixel entry: byte in row (8 bits) | ring row (4 bits) << 8 | dark << 12
        e0 = ((rec & 63u) << 2) | ((rec + 0x300u) & 0xF00u);
        mb = rec & 0x80808080u;
        md = (rec << 1) & 0x80808080u;
      }
      const uint32_t mine = __popc(mb) + __popc(md);
      const uint32_t incl = wave_incl_scan(mine);
      uint32_t total = __builtin_amdgcn_readlane(incl, 63), take = 64;
      if (total > (uint32_t)kSPCap) {              // uniform, rare: the first 32 records only (at most 256 entries)
        take = 32;
        total = __builtin_amdgcn_readlane(incl, 31);
        if (lane >= 32) mb = md = 0u;
      }
      if (nP + total > (uint32_t)kSPCap) score_pending();
      uint32_t k = nP + incl - mine;
      if (mb & 0x80u) sm.p[k++] = (uint16_t)e0;
      if (mb & 0x8000u) sm.p[k++] = (uint16_t)(e0 + 1);
      if (mb & 0x800000u) sm.p[k++] = (uint16_t)(e0 + 2);
      if (mb & 0x80000000u) sm.p[k++] = (uint16_t)(e0 + 3);
      e0 |= 0x1000u;
      if (md & 0x80u) sm.p[k++] = (uint16_t)e0;
      if (md & 0x8000u) sm.p[k++] = (uint16_t)(e0 + 1);
      if (md & 0x800000u) sm.p[k++] = (uint16_t)(e0 + 2);
      if (md & 0x80000000u) sm.p[k++] = (uint16_t)(e0 + 3);
      nP += total;
      c0 += take;
    }
    score_pending();
    nG = 0;
    // ---- NMS of rel rows [nms_lo, s - 1]: all their neighbours' scores are final
    const int hi = s - 1;
    if (!overflow) {
      for (uint32_t i0 = 0; i0 < nH; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool keep = false;
        uint32_t key = 0, krr = 0, c = 0;
        if (i < nH) {
          const uint32_t he = sm.h[i];
          const uint32_t xl = he & 255u, row = (he >> 8) & 15u;                 // byte inside the row = 4 * lane + j
          // the list holds rel rows nms_lo .. s, fewer than RK of them: the ring row identifies the rel row
          const int rr = nms_lo + (int)((row - (uint32_t)(nms_lo + 3)) & (uint32_t)(RK - 1));
          const uint8_t* qc = scb + (row << 8) + xl;
          const uint8_t* qu = scb + (((row + RK - 1) & (RK - 1)) << 8) + xl;
          const uint8_t* qd = scb + (((row + 1) & (RK - 1)) << 8) + xl;
          c = qc[0];
          const int x = xs + (int)xl;
          keep = rr >= nms_lo && rr <= hi && rr >= o_lo && rr < o_hi && x >= ox_lo && x < ox_hi;
          keep = keep && c > qc[-1] && c > qc[1] && c > qu[-1] && c > qu[0] && c > qu[1] && c > qd[-1] && c > qd[0] &&
                 c > qd[1];
          key = xl;
          krr = (uint32_t)rr;
        }
        emit(keep, key, krr, c);
      }
    } else {
      for (int rr = nms_lo; rr <= hi; rr++) {
        if (rr < o_lo || rr >= o_hi) continue;
        const int row = (rr + 3) & (RK - 1);
        const int w = (row << 6) + lane, wu = (((row + RK - 1) & (RK - 1)) << 6) + lane, wd = (((row + 1) & (RK - 1)) << 6) + lane;
        const uint32_t C = scw[w];
        if (__ballot(C != 0u) == 0ull) continue;
        // left / right dwords by DPP: the window's edge lanes' outer neighbours are never output (om = 0 there)
        const uint32_t U = scw[wu], D = scw[wd];
#define SHR1(v_) __builtin_amdgcn_update_dpp(0u, (v_), 0x138, 0xf, 0xf, true)
#define SHL1(v_) __builtin_amdgcn_update_dpp(0u, (v_), 0x130, 0xf, 0xf, true)
        uint32_t keep = __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(C, SHR1(C), 3), 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(SHL1(C), C, 1), 0);
        keep &= __builtin_amdgcn_lerp(C, ~U, 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(U, SHR1(U), 3), 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(SHL1(U), U, 1), 0);
        keep &= __builtin_amdgcn_lerp(C, ~D, 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(D, SHR1(D), 3), 0);
        keep &= __builtin_amdgcn_lerp(C, ~__builtin_amdgcn_alignbyte(SHL1(D), D, 1), 0);
#undef SHR1
#undef SHL1
        keep &= om;
#pragma unroll
        for (int j = 0; j < 4; j++) emit((keep >> (8 * j + 7)) & 1u, (uint32_t)(4 * lane + j), (uint32_t)rr, (C >> (8 * j)) & 255u);
      }
    }
    // the list restarts with the scored corners of row s (their NMS needs row s + 1)
    nH = 0;
    {
      const uint32_t row = (uint32_t)(s + 3) & (RK - 1);
      const uint32_t C = scw[(row << 6) + lane];
      if (__ballot(C != 0u) != 0ull)               // most rows hold no scored corner at a high threshold
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool hit = ((C >> (8 * j)) & 255u) != 0u;
        const unsigned long long bal = __ballot(hit);
        if (bal) {
          const uint32_t cnt = (uint32_t)__popcll(bal);
          // a row holds at most 256 corners; the list keeps what fits and the next flush falls back to the dense NMS
          if (hit && nH + mbcnt64(bal) < (uint32_t)kSHCap) sm.h[nH + mbcnt64(bal)] = (uint16_t)((4u * lane + j) | (row << 8));
          nH += cnt;
        }
      }
    }
    MSF_WAVE_SYNC();
    nms_lo = s;
    last_flush = s;
  };

  // warm-up: pixel rows y0 - 3 .. y0 + 2 (ring indices 0 .. 5), then the queue holds rows y0 + 3 .. y0 + 6
  PUT_ROW(0, w0_); PUT_ROW(1, w1_); PUT_ROW(2, w2_); PUT_ROW(3, w3_); PUT_ROW(4, w4_); PUT_ROW(5, w5_);
  // resize: rows R0 = y0 + 1 (ring index 4) and R0 + 1 (index 5) are in; the output row (if any) between them goes out
  uint32_t hp[4] = {0, 0, 0, 0};                  // sums of the newest row handled
  if (RESIZE) {
    uint32_t hu[4];
    RZ_HSUM(4, hu);
    RZ_HSUM(5, hp);
    RZ_EMIT(4, hu, hp);                           // upper row R0 = table entry 4
  }
  // One group = four rel rows s .. s+3, in three straight-line parts so that the four rows' dependent chains overlap
  // (a wave has 4 x the instruction-level parallelism of one row at a time; the kernel is latency-bound):
  //  1. the four queued rows (row + 3 of each step) go to the ring and their queue slots are refilled four rows ahead
  //     -- each queue register is named statically: a rotating queue would make every step wait for the newest load;
  //  2. the cardinal prefilters of the four rows (ring indices s_, s_ + 3, s_ + 6; see fast_tile for the SWAR form):
  //     the row three below comes from the register just stored, the others from the ring;
  //  3. the four record appends.
  // The tests against the pixel three rows ABOVE are not computed: row y - 3 has already compared itself with row y (its
  // pixel three rows below), and "U brighter than C by more than tau" is "C darker than U by more than tau" seen from
  // row y - 3.  With bu_ / nu_ = the "below is brighter" / "below is not darker" bits of row y - 3 (bit 7 of each byte):
  //   above brighter  [U > C + tau]  is taken as  ~nu_ = [U >= C + tau]   (a superset: the equality case passes too)
  //   above not darker               is taken as  ~bu_ = [U >= C - tau]   (dark-up = bu_ = [U < C - tau], FAST's strict test)
  // Both keep the prefilter a necessary condition for a corner of score > tau (exact scores decide), and a row costs 9
  // v_lerp_u8 instead of 12 and no ring read of the row above.  b0o_ / n0o_: this row's bits for the row three below.
#define STREAM_PRE(C_, D_, bu_, nu_, vmr_, cb_, cd_, b0o_, n0o_)                                                        \
  do {                                                                                                                 \
    const uint32_t Lf_ = __builtin_amdgcn_update_dpp(0u, C_, 0x138, 0xf, 0xf, true); /* wave_shr:1: lane i <- i-1 */   \
    const uint32_t Rt_ = __builtin_amdgcn_update_dpp(0u, C_, 0x130, 0xf, 0xf, true); /* wave_shl:1: lane i <- i+1 */   \
    const uint32_t W3_ = __builtin_amdgcn_alignbyte(C_, Lf_, 1);                                                       \
    const uint32_t E3_ = __builtin_amdgcn_alignbyte(Rt_, C_, 3);                                                       \
    const uint32_t nC_ = ~(C_);                                                                                        \
    const uint32_t l0_ = __builtin_amdgcn_lerp(D_, nC_, 0), l4_ = __builtin_amdgcn_lerp(E3_, nC_, 0);                  \
    const uint32_t l12_ = __builtin_amdgcn_lerp(W3_, nC_, 0);                                                          \
    const uint32_t b0_ = __builtin_amdgcn_lerp(l0_, lerp_bright, 0), b4_ = __builtin_amdgcn_lerp(l4_, lerp_bright, 0);     \
    const uint32_t b12_ = __builtin_amdgcn_lerp(l12_, lerp_bright, 0);                                                 \
    const uint32_t n0_ = __builtin_amdgcn_lerp(l0_, lerp_not_dark, 0), n4_ = __builtin_amdgcn_lerp(l4_, lerp_not_dark, 0); \
    const uint32_t n12_ = __builtin_amdgcn_lerp(l12_, lerp_not_dark, 0);                                               \
    cb_ = ((b0_ | ~(nu_)) & (b4_ | b12_)) & (vmr_);                                                                    \
    cd_ = ~((n0_ & ~(bu_)) | (n4_ & n12_)) & (vmr_);                                                                   \
    b0o_ = b0_;                                                                                                        \
    n0o_ = n0_;                                                                                                        \
  } while (0)
  // the same for a row whose record is never made (the three rows above the strip's first): only its bits for the row below
#define STREAM_PRE_DOWN(C_, D_, b0o_, n0o_)                                                                            \
  do {                                                                                                                 \
    const uint32_t l0_ = __builtin_amdgcn_lerp(D_, ~(C_), 0);                                                          \
    b0o_ = __builtin_amdgcn_lerp(l0_, lerp_bright, 0);                                                                 \
    n0o_ = __builtin_amdgcn_lerp(l0_, lerp_not_dark, 0);                                                               \
  } while (0)
#define STREAM_APPEND(s_, cb_, cd_)                                                                                    \
  do {                                                                                                                 \
    const bool has_ = ((cb_) | (cd_)) != 0u;                                                                           \
    const unsigned long long bal_ = __ballot(has_);                                                                    \
    if (bal_) {                                                                                                        \
      /* cb_ / cd_ hold bit 7 of each byte only: the flags stay where they are, lane and row go into the free low bits */ \
      if (has_) sm.g[nG + mbcnt64(bal_)] = (cb_) | ((cd_) >> 1) | lane_rec | ((((uint32_t)(s_) & 63u) << 8) | (((uint32_t)(s_) >> 6) << 16)); \
      nG += (uint32_t)__popcll(bal_);                                                                                  \
    }                                                                                                                  \
  } while (0)
  const uint32_t lane_rec = (uint32_t)lane;
  static_assert(kWkMaxRows + 12 < 256, "rel rows: 8 bits in a buffered corner, 4 x 64 emit entries");
  static_assert(RK == 16, "the group body below exists in four copies, one per position of the group in the 16-row ring");
  // The ring slot of a row is (ring index) mod 16 and a group starts at a multiple of 4: the body is instantiated for the
  // four values of (s mod 16), so that every slot is a compile-time constant and every LDS access of the group is
  // lane base + immediate offset (18 vector instructions per group went into ring addresses).  P_ = s mod 16.
  // rel rows -3 .. -1 (ring indices 0 .. 2) against the rows three below them (ring indices 3 .. 5)
  uint32_t pb1, pn1, pb2, pn2, pb3, pn3;
  STREAM_PRE_DOWN(w0_, w3_, pb1, pn1);
  STREAM_PRE_DOWN(w1_, w4_, pb2, pn2);
  STREAM_PRE_DOWN(w2_, w5_, pb3, pn3);
  auto group = [&](auto ph_, const int s) {
    constexpr int P_ = decltype(ph_)::value * 4;
    if (RESIZE && ((s + 5) >> 6) != em_k) {          // uniform, once per 64 rows: the window of the emit table moves on
      em_k++;
      em_a = em_b;
      em_b = em_k == 1 ? em2 : em_k == 2 ? em3 : 0u;
    }
    const uint32_t d0 = q0, d1 = q1, d2 = q2, d3 = q3;
    PUT_ROW(P_ + 6, d0); q0 = LOAD_ROW(y0 + s + 7);
    PUT_ROW(P_ + 7, d1); q1 = LOAD_ROW(y0 + s + 8);
    PUT_ROW(P_ + 8, d2); q2 = LOAD_ROW(y0 + s + 9);
    PUT_ROW(P_ + 9, d3); q3 = LOAD_ROW(y0 + s + 10);
    const uint32_t c0 = pxw[((P_ + 3) & (RK - 1)) * 64 + lane];
    const uint32_t c1 = pxw[((P_ + 4) & (RK - 1)) * 64 + lane], c2 = pxw[((P_ + 5) & (RK - 1)) * 64 + lane];
    if (RESIZE) {
      // the four rows just stored are source rows y0 + s + 3 .. y0 + s + 6 = R0 + s + 2 .. (table entries s + 6 ..);
      // an output row between rows (y - 1, y) has the UPPER row's entry: s + 5 .. s + 8
      uint32_t ha[4], hb[4], hc[4], hd[4];
      RZ_HSUM(P_ + 6, ha);
      RZ_HSUM(P_ + 7, hb);
      RZ_HSUM(P_ + 8, hc);
      RZ_HSUM(P_ + 9, hd);
      RZ_EMIT(s + 5, hp, ha);
      RZ_EMIT(s + 6, ha, hb);
      RZ_EMIT(s + 7, hb, hc);
      RZ_EMIT(s + 8, hc, hd);
#pragma unroll
      for (int k = 0; k < 4; k++) hp[k] = hd[k];
    }
    uint32_t cb0, cd0, cb1, cd1, cb2, cd2, cb3, cd3;
    // rows outside the strip's scored rows [s_lo, s_hi] only occur in its first and last groups: one test per group
    uint32_t vm0 = vm, vm1 = vm, vm2 = vm, vm3 = vm;
    if (s < s_lo || s + 3 > s_hi) {
      vm0 = (s >= s_lo && s <= s_hi) ? vm : 0u;
      vm1 = (s + 1 >= s_lo && s + 1 <= s_hi) ? vm : 0u;
      vm2 = (s + 2 >= s_lo && s + 2 <= s_hi) ? vm : 0u;
      vm3 = (s + 3 >= s_lo && s + 3 <= s_hi) ? vm : 0u;
    }
    // (pb1, pn1) .. (pb3, pn3): the "below" bits of rel rows s - 3 .. s - 1; this group's rows s + 1 .. s + 3 replace them
    uint32_t rb0, rn0;
    STREAM_PRE(c0, d0, pb1, pn1, vm0, cb0, cd0, rb0, rn0);
    STREAM_PRE(c1, d1, pb2, pn2, vm1, cb1, cd1, pb1, pn1);
    STREAM_PRE(c2, d2, pb3, pn3, vm2, cb2, cd2, pb2, pn2);
    STREAM_PRE(d0, d3, rb0, rn0, vm3, cb3, cd3, pb3, pn3);   // the centre row of step s + 3 is the row stored first in this group
    STREAM_APPEND(s, cb0, cd0);
    STREAM_APPEND(s + 1, cb1, cd1);
    STREAM_APPEND(s + 2, cb2, cd2);
    STREAM_APPEND(s + 3, cb3, cd3);
  };
  for (int s = 0; s <= r_last; s += 4) {
    switch ((s >> 2) & 3) {
      case 0: group(std::integral_constant<int, 0>{}, s); break;
      case 1: group(std::integral_constant<int, 1>{}, s); break;
      case 2: group(std::integral_constant<int, 2>{}, s); break;
      default: group(std::integral_constant<int, 3>{}, s); break;
    }
    const int sl = min(s + 3, r_last);
    // flushes happen between groups of four steps: at most kSGFlush + 4 x 64 records wait (kSGCap), at most 8 rows
    if (do_fast && (nG > kSGFlush || sl - last_flush >= kFlushRows || sl == r_last)) {
      MSF_WAVE_SYNC();
      flush(sl);
    }
  }
#undef STREAM_PRE
#undef STREAM_PRE_DOWN
#undef STREAM_APPEND
  flush_out();
#undef LOAD_ROW
#undef PUT_ROW
#undef RZ_HSUM
#undef RZ_EMIT
  }
  if (count_me || RESIZE) {
    // Every store of this strip has left the wave's queue before the strip counts as done: the write-through stores of
    // level l + 1's pixels are then in memory (what the next level's units wait for) and the histogram adds have been
    // performed (agent-scope atomics both sides).  Inline asm: the compiler may not drop or move this wait.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (count_me) atomicAdd(&qs[kQDone], 1u);
      if (RESIZE) atomicAdd(&qs[kQAll], 1u);
    }
  }
}

// ------------------------------------------------------------------ output-sensitive FAST: threshold estimate, check, redo
// retainBest(2N) (KeyPointsFilter) keeps, per level, the strict maxima whose FAST score reaches the
// (2N)-th largest one: a few hundred of the tens of thousands a textured frame has.  Every maximum with score >= tau
// is found exactly by fast_tile(tau), so if at least 2N of them exist the kept set is already complete and nothing
// below tau was ever needed.  tau_unit (a unit of the walker launch) picks the first tau per (frame, level) -- predicted
// from the level above or from exact scores on a sparse sample of the level --, k_fast_check verifies the count afterwards and queues the (frame, level)s that fell short for a dense
// (tau = fastThreshold) second pass by k_fast_redo.  Whatever tau is picked, the result is the dense one bit for bit.
constexpr int kTauMinHits = 24;       // sample hits the estimate must rest on
constexpr int kTauOversample = 10;    // estimated pixels with score >= tau per key point to keep: a strict maximum stands
                                      // for 4-5 pixels of its cluster at these scores, so about twice the 2N needed

__device__ __forceinline__ int fast_score_px(const uint8_t* p, int pitch) {
  const int off[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
  int q[16];
#pragma unroll
  for (int k = 0; k < 16; k++) q[k] = (int)p[off[k][1] * pitch + off[k][0]];
  int mn3[16], mx3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    mn3[k] = min(min(q[k], q[(k + 1) & 15]), q[(k + 2) & 15]);
    mx3[k] = max(max(q[k], q[(k + 1) & 15]), q[(k + 2) & 15]);
  }
  int Wb = -1, Wd = 1000;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    Wb = max(Wb, min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]));
    Wd = min(Wd, max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]));
  }
  const int v = p[0];
  const int A = max(Wb - v, v - Wd);
  return A > kFastT ? A - 1 : 0;
}

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// The threshold unit of (frame fi, level l): one wave.  Sampled pixels: rows samp_sy apart with a hashed jitter, columns
// samp_sx apart -- odd, hashed phase per row, so a periodic texture is not aliased.  Only the upper tail of the score
// distribution matters, so a sample first takes the cardinal test at kTauPre (5 byte loads; every pixel with score >=
// kTauPre passes it) and only the survivors, compacted in LDS, get the exact score: the histogram is exact from kTauPre
// upwards.  tau = the largest multiple of 4 with enough sample hits at or above it, fastThreshold (dense) if there is none.
constexpr int kTauPre = 40, kTauPreHigh = 64;
constexpr int kTauListCap = 2048;     // prefilter survivors kept per (frame, level); the rest are dropped (fewer hits: a
                                      // lower, still valid, tau)
struct TauSmem {
  uint32_t hist[kTauBins];
  uint32_t nlist, pad[3];
  uint32_t list[kTauListCap];
};
static_assert(sizeof(TauSmem) <= sizeof(StreamSmem), "the threshold unit works in the walker's LDS block");

__device__ __forceinline__ void tau_unit(TauSmem& T, const OrbGeometry& g, const FrameSrc& src, const uint8_t* pyr,
                                         uint32_t* tau, uint32_t* tau_first, uint32_t* redo_cnt, uint32_t* redo_list,
                                         uint32_t* qstat, uint32_t* status, uint32_t* abort_word, int force_tau,
                                         int predict_pct, int chain, const int l, const int fi) {
  const int lane = threadIdx.x, slot = src.slot0 + fi, idx = slot * kOrbLevels + l;
  const OrbLevelInfo L = g.lv[l];
  uint32_t* const qs = qstat + (size_t)idx * kQStat;
  int tv = kFastT, pre_used = kTauPre, predicted = 0;
  if (chain && l > 0) {
    // level l exists once every strip of level l - 1 has counted itself done (walk_strip's last lines)
    const OrbLevelInfo Lp = g.lv[l - 1];
    const uint32_t* pq = qstat + (size_t)(idx - 1) * kQStat;
    const int n_prev = Lp.wk_nx * Lp.wk_ny, qa = (n_prev + 3) >> 2;
    uint32_t pv = ld_agent(&pq[lane]);
    uint32_t px = ld_agent(&pq[kTauBins + (lane & 3)]);
    bool ready = false;
    for (uint32_t it = 0; it < kSpinMax; it++) {
      ready = (uint32_t)__builtin_amdgcn_readlane((int)px, kQAll - kTauBins) >= (uint32_t)n_prev;
      if (ready || ld_agent(abort_word) != 0u) break;     // uniform
      __builtin_amdgcn_s_sleep(20);
      pv = ld_agent(&pq[lane]);
      px = ld_agent(&pq[kTauBins + (lane & 3)]);
    }
    if (!ready) {
      unit_stall(status, slot, abort_word, lane);
      return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the sampler below reads level l's pixels
    // Prediction from the level above (predict_pct > 0).  The sampled quarter of level l - 1 left the histogram of its
    // exact corners; level l shows the same scene 1.2 x smaller, so the first threshold of level l is the largest
    // multiple of 4 at which level l - 1 -- scaled from the quarter to the level -- still holds predict_pct % of the
    // corner DENSITY level l needs for its 2N.  It only steers the quarter of level l (the rest refines from that
    // quarter's own corners) and k_fast_check + the dense redo catch an overshoot, so whatever comes out here the result
    // is the dense one.  No usable histogram (a dense level above, fewer than two sampled strips, too few corners): the
    // sampler below runs.
    if (predict_pct > 0 && force_tau == 0 && L.tiles_x > 0) {
      const uint32_t done = (uint32_t)__builtin_amdgcn_readlane((int)px, kQDone - kTauBins);
      const uint32_t t_q = (uint32_t)__builtin_amdgcn_readlane((int)px, kQFirst - kTauBins) & ~kQReady;
      uint32_t c = pv;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_down(c, o);
        if (lane + o < 64) c += up;
      }
      if (qa >= kTau2MinStrips && done >= (uint32_t)qa && t_q > (uint32_t)kFastT) {
        const long long a_prev = (long long)(Lp.w - 2 * kEdge) * (Lp.h - 2 * kEdge), a_cur = (long long)(L.w - 2 * kEdge) * (L.h - 2 * kEdge);
        // corners the QUARTER must hold at or above t: predict_pct % x 2N_l x (area above / area here) x (quarter / level)
        const long long need = ((long long)predict_pct * 2 * L.quota * a_prev * qa + 100ll * a_cur * n_prev - 1) / (100ll * a_cur * n_prev);
        const unsigned long long ok = __ballot((long long)c >= need && 4u * (uint32_t)lane >= t_q);
        if (ok) predicted = 4 * (63 - __builtin_clzll(ok));
      }
    }
  }
  if (predicted > kFastT) {
    tv = predicted;
  } else if (force_tau > 0) {
    tv = force_tau;
  } else if (L.samp_rows > 0) {
    if (lane < kTauBins) T.hist[lane] = 0;
    if (lane == 0) T.nlist = 0;
    MSF_WAVE_SYNC();
    int pitch;
    const uint8_t* img = level_ptr(g, src, pyr, fi, l, &pitch);
    const int rh = L.h - 2 * kEdge;
    // A sample site is a run of 4 adjacent pixels (one aligned dword): the cardinal test then works on 4 px at once
    // from 5 dword loads (the SWAR form of fast_tile's prefilter at kTauPre), a quarter of the loads per pixel.
    // Four sampled rows per iteration, so that their 20 loads are in flight together.
    // The test runs at kTauPreHigh first; textures without enough strong corners (few survivors: the estimate would
    // end below kTauPreHigh anyway) are sampled again at kTauPre.
    const uint32_t factor_px = (uint32_t)(L.samp_sx * L.samp_sy);
    uint32_t need_hits = ((uint32_t)kTauOversample * 2u * (uint32_t)L.quota + factor_px - 1u) / factor_px;
    need_hits = need_hits < (uint32_t)kTauMinHits ? (uint32_t)kTauMinHits : need_hits;
    int pre = kTauPreHigh;
    for (int attempt = 0; attempt < 2; attempt++) {
      const uint32_t lb = 0x01010101u * (uint32_t)(128 - pre / 2), lnd = 0x01010101u * (uint32_t)(255 - (254 - pre) / 2);
      const int x_first = (kEdge + 3) & ~3;                                  // first dword fully inside [31, w - 31)
      const int n_dw = (L.w - kEdge - x_first) >> 2;                         // dwords fully inside
      for (int j0 = 0; j0 < L.samp_rows; j0 += 4) {
        for (int k = lane; k < L.samp_cols; k += 64) {
          uint32_t Cv[4], Lv[4], Rv[4], Uv[4], Dv[4], pos[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int j = min(j0 + u, L.samp_rows - 1);
            const uint32_t hj = hash_u32((uint32_t)(l * 4099 + j) * 2654435761u + 12345u);
            int y = j * L.samp_sy + (int)(hj % (uint32_t)L.samp_sy);
            y = kEdge + (y < rh ? y : rh - 1);
            int d = k * L.samp_sx + (int)((hj >> 16) % (uint32_t)L.samp_sx);   // dword index of the run
            d = d < n_dw ? d : n_dw - 1;
            const int x = x_first + 4 * d;
            const uint8_t* p = img + (long long)y * pitch + x;
            Cv[u] = *reinterpret_cast<const uint32_t*>(p);
            Lv[u] = *reinterpret_cast<const uint32_t*>(p - 4);
            Rv[u] = *reinterpret_cast<const uint32_t*>(p + 4);
            Uv[u] = *reinterpret_cast<const uint32_t*>(p - 3 * pitch);
            Dv[u] = *reinterpret_cast<const uint32_t*>(p + 3 * pitch);
            pos[u] = ((uint32_t)y << 16) | (uint32_t)x;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t W3 = __builtin_amdgcn_alignbyte(Cv[u], Lv[u], 1), E3 = __builtin_amdgcn_alignbyte(Rv[u], Cv[u], 3);
            const uint32_t nC = ~Cv[u];
            const uint32_t l0 = __builtin_amdgcn_lerp(Dv[u], nC, 0), l4 = __builtin_amdgcn_lerp(E3, nC, 0);
            const uint32_t l8 = __builtin_amdgcn_lerp(Uv[u], nC, 0), l12 = __builtin_amdgcn_lerp(W3, nC, 0);
            const uint32_t cb = (__builtin_amdgcn_lerp(l0, lb, 0) | __builtin_amdgcn_lerp(l8, lb, 0)) &
                                (__builtin_amdgcn_lerp(l4, lb, 0) | __builtin_amdgcn_lerp(l12, lb, 0));
            const uint32_t cd = ~((__builtin_amdgcn_lerp(l0, lnd, 0) & __builtin_amdgcn_lerp(l8, lnd, 0)) |
                                  (__builtin_amdgcn_lerp(l4, lnd, 0) & __builtin_amdgcn_lerp(l12, lnd, 0)));
            uint32_t m = (cb | cd) & 0x80808080u;
            if (j0 + u >= L.samp_rows) m = 0u;
            if (m) {
              const uint32_t idx0 = atomicAdd(&T.nlist, (uint32_t)__popc(m));
              uint32_t q = idx0;
#pragma unroll
              for (int b4 = 0; b4 < 4; b4++)
                if ((m >> (8 * b4 + 7)) & 1u) {
                  if (q < (uint32_t)kTauListCap) T.list[q] = pos[u] + (uint32_t)b4;
                  q++;
                }
            }
          }
        }
      }
      MSF_WAVE_SYNC();
      // about one survivor of the cardinal test in four or five scores above the test's threshold
      if (attempt == 1 || T.nlist >= 6u * need_hits) break;   // uniform
      MSF_WAVE_SYNC();
      if (lane == 0) T.nlist = 0;
      pre = kTauPre;
      MSF_WAVE_SYNC();
    }
    pre_used = pre;
    const uint32_t n = min(T.nlist, (uint32_t)kTauListCap);
    for (uint32_t i = lane; i < n; i += 64) {
      const uint32_t e = T.list[i];
      const int sc = fast_score_px(img + (long long)(e >> 16) * pitch + (e & 0xFFFFu), pitch);
      if (sc >= pre_used) atomicAdd(&T.hist[sc >> 2], 1u);
    }
    MSF_WAVE_SYNC();
    {
      // suffix sums over the 64 bins: c = hits with score >= 4 * lane
      uint32_t c = T.hist[lane];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_down(c, o);
        if (lane + o < 64) c += up;
      }
      const unsigned long long ok = __ballot(c >= need_hits && 4 * lane >= pre_used);
      const int top = ok ? 63 - __builtin_clzll(ok) : 0;     // largest qualifying bin
      tv = ok ? 4 * top : kFastT;                            // too few strong corners: dense
    }
    MSF_WAVE_SYNC();                                         // the LDS block is the next unit's
  }
  if (lane == 0) {
    tau[idx] = (uint32_t)tv;
    tau_first[idx] = (uint32_t)tv;
    // nothing to gain from a threshold: straight to the dense pass
    if (tv <= kFastT && L.tiles_x > 0) redo_list[atomicAdd(redo_cnt, 1u)] = (uint32_t)(fi * kOrbLevels + l);
    // the strips of this (frame, level) start from here
    __hip_atomic_store(&qs[kQFirst], kQReady | (uint32_t)tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The walker launch: see "the one-launch form" above walk_strip.  resize_mask bit l: the strips of level l also make
// level l + 1.  One wave per workgroup, one unit per workgroup.
__global__ __launch_bounds__(64) void k_walk(OrbGeometry g, FrameSrc src, uint8_t* pyr, const uint32_t* __restrict__ tab,
                                             uint32_t* tau, uint32_t* tau_first, uint32_t* qstat, uint32_t* cand_cnt,
                                             uint32_t* cand_key, uint8_t* cand_sc, const uint2* cmap,
                                             uint32_t* redo_cnt, uint32_t* redo_list,
                                             uint32_t* status, uint32_t* abort_word, int l_lo, int l_hi, int n_frames,
                                             int margin_pct, int dyn, int force_tau, int predict_pct, int chain,
                                             int resize_mask, int test_stall_frame) {
  __shared__ StreamSmem sm;
  const int xcd = (int)(blockIdx.x & 7u);
  int r = (int)(blockIdx.x >> 3);                // unit of this XCD
  const int nf8 = n_frames >> 3, nrem = n_frames & 7;
  const int nfx = nf8 + (xcd < nrem ? 1 : 0), f0 = xcd * nf8 + (xcd < nrem ? xcd : nrem);
  int l = l_lo, S = 0;
  for (;; l++) {
    if (l > l_hi) return;                        // surplus workgroups of an XCD with one frame fewer
    S = g.lv[l].wk_nx * g.lv[l].wk_ny;
    const int units = nfx * (1 + S);
    if (r < units) break;
    r -= units;
  }
  if (r < nfx) {
    // test hook (MSF_TEST_HOOKS=1 + MSF_ORB_TEST_STALL_FRAME=f, never set otherwise): the threshold of (frame f, level 3)
    // is never published, so that the bounded waits, the abort word and the per-frame error flag can be exercised
    if (f0 + r == test_stall_frame && l == 3) return;
    tau_unit(*reinterpret_cast<TauSmem*>(&sm), g, src, pyr, tau, tau_first, redo_cnt, redo_list, qstat, status, abort_word,
             force_tau, predict_pct, chain, l, f0 + r);
    return;
  }
  r -= nfx;
  const int Sq = (S + 3) >> 2, Sr = S - Sq;
  int fi, ts;
  if (r < nfx * Sq) {
    fi = f0 + r / Sq;
    ts = 4 * (r % Sq);
  } else {
    r -= nfx * Sq;
    fi = f0 + r / Sr;
    const int ip = r % Sr;
    ts = 4 * (ip / 3) + ip % 3 + 1;
  }
  if ((resize_mask >> l) & 1)
    walk_strip<true>(sm, g, src, pyr, tab, qstat, cand_cnt, cand_key, cand_sc, cmap, status, abort_word, margin_pct, dyn, chain, l, fi, ts);
  else
    walk_strip<false>(sm, g, src, pyr, tab, qstat, cand_cnt, cand_key, cand_sc, cmap, status, abort_word, margin_pct, dyn, chain, l, fi, ts);
}

// After the walker: tau[idx] is the largest threshold any strip of the (frame, level) ran at, and every strict maximum
// with a score at or above it is in the candidate list (strips that ran lower also left smaller ones).  If fewer than 2N
// reach it, retainBest(2N) would cut below what was searched completely: the level is queued for the dense pass (its
// candidate list restarts from empty).  One wave per (frame, level).
__global__ __launch_bounds__(64) void k_fast_check(OrbGeometry g, int slot0, int n_frames, uint32_t* tau, uint32_t* tau_first,
                                                   uint32_t* cand_cnt, const uint8_t* cand_sc, const uint2* cmap,
                                                   uint32_t* redo_cnt, uint32_t* redo_list, const uint32_t* qstat) {
  const int fi = blockIdx.x / kOrbLevels, l = blockIdx.x - fi * kOrbLevels, lane = threadIdx.x;
  if (fi >= n_frames || l >= g.nlevels) return;
  const int idx = (slot0 + fi) * kOrbLevels + l;
  if (tau[idx] <= (uint32_t)kFastT) return;
  const uint32_t T = max(tau[idx], qstat[(size_t)idx * kQStat + kQTau]);   // the largest threshold any strip ran at
  const OrbLevelInfo L = g.lv[l];
  const uint2 cm = cmap[idx];
  // a primary list that overflowed is incomplete: the level takes the dense pass (with a full-capacity block) like one
  // that holds too few strong corners
  const bool over = cand_cnt[idx] > cm.y;
  const uint32_t n = min(cand_cnt[idx], cm.y);
  const uint8_t* scs = cand_sc + cm.x;
  uint32_t found = 0;
  for (uint32_t i = 16u * lane; i < n; i += 16u * 64u) {
    const uint4 v = *reinterpret_cast<const uint4*>(scs + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int b = 0; b < 16; b++)
      if (i + b < n && ((w[b >> 2] >> (8 * (b & 3))) & 255u) >= T) found++;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) found += __shfl_xor(found, o);
  if (lane == 0) {
    if (over || found < 2u * (uint32_t)L.quota) {
      tau[idx] = kFastT;
      cand_cnt[idx] = 0;
      redo_list[atomicAdd(redo_cnt, 1u)] = (uint32_t)(fi * kOrbLevels + l);
    } else {
      tau[idx] = T;           // the threshold the list is complete from (MSF_DBG_FAST_TAU reports both)
      tau_first[idx] = T;
    }
  }
}

// Dense second pass over the queued (frame, level)s: fixed grid, unit u = (queue entry, tile index); tile indices past
// the level's tile count are skipped (levels differ in size; the queue is short or empty in practice).
__global__ __launch_bounds__(kFastThreads) void k_fast_redo(OrbGeometry g, FrameSrc src, const uint8_t* pyr,
                                                            const uint32_t* __restrict__ redo_cnt,
                                                            const uint32_t* __restrict__ redo_list, int max_tiles,
                                                            uint32_t* cand_cnt, uint32_t* cand_key, uint8_t* cand_sc,
                                                            const uint2* cmap) {
  __shared__ FastSmem sm;
  const uint32_t n = *redo_cnt;
  const unsigned long long units = (unsigned long long)n * (unsigned)max_tiles;
  // workgroups b, b + 8, ... share an XCD (L2): they walk one contiguous eighth of the units together
  const unsigned long long per_xcd = (units + 7ull) >> 3;
  const unsigned long long u_begin = (blockIdx.x & 7u) * per_xcd;
  const unsigned long long u_end = u_begin + per_xcd < units ? u_begin + per_xcd : units;
  for (unsigned long long u = u_begin + (blockIdx.x >> 3); u < u_end; u += (gridDim.x >> 3)) {
    const uint32_t item = redo_list[u / (unsigned)max_tiles];
    const int t = (int)(u % (unsigned)max_tiles);
    const int fi = (int)(item / kOrbLevels), l = (int)(item % kOrbLevels);
    if (t >= g.lv[l].tiles_x * g.lv[l].tiles_y) continue;   // uniform
    fast_tile(g, src, pyr, fi, l, t, kFastT, cand_cnt, cand_key, cand_sc, cmap, sm);
    __syncthreads();
  }
}

// After the dense second pass: a redone level whose candidates did not fit its primary list (a frame of noise: tens of
// thousands of maxima; a smooth frame's dense list is short) moves to a pool region of full capacity and is queued once
// more.  One thread per entry of the first queue.  Pool exhausted: the frame is flagged (MSF_ERR_CAPACITY), never a short list.
__global__ __launch_bounds__(256) void k_redo_overflow(OrbGeometry g, int slot0, uint32_t* cand_cnt, uint2* cmap,
                                                       uint32_t* pool_cnt, uint32_t* status, const uint32_t* redo_cnt,
                                                       const uint32_t* redo_list, uint32_t* redo2_cnt, uint32_t* redo2_list) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= *redo_cnt) return;
  const uint32_t item = redo_list[i];
  const int fi = (int)(item / kOrbLevels), l = (int)(item % kOrbLevels);
  const int idx = (slot0 + fi) * kOrbLevels + l;
  if (cand_cnt[idx] <= cmap[idx].y) return;
  if ((uint32_t)g.lv[l].cand_cap > cmap[idx].y && cand_take_block(g, cmap, pool_cnt, idx, l)) {
    cand_cnt[idx] = 0;
    redo2_list[atomicAdd(redo2_cnt, 1u)] = item;
  } else {
    atomicOr(&status[slot0 + fi], kStatusOverflow);
  }
}

// ------------------------------------------------------------------ K5+K6: retainBest(2N) by FAST score, Harris response
// HarrisResponses (orb.cpp), blockSize 7, k = 0.04, on the unblurred level.
// One lane per candidate.  The 9 x 9 neighbourhood comes in as 27 aligned dwords (3 per row; scattered byte loads --
// ~190 per candidate -- made the texture addresser the bottleneck: 0.67 ms of this stage per 2048 frames), and the
// Sobel sums are separable dot products on bytes biased by 128 (differences and the 1-2-1 sums of differences are
// unaffected by the bias):  G[row][j] = p[j+1] - p[j-1],  H[row][j] = p[j-1] + 2 p[j] + p[j+1] - 512,
//   Ix[r][j] = G[r-1][j] + 2 G[r][j] + G[r+1][j],   Iy[r][j] = H[r+1][j] - H[r-1][j]   (all int32, exact).
__device__ __forceinline__ float harris_at(const uint8_t* img, int step, int x0, int y0) {
  const int xb = (x0 - 4) & ~3;
  const uint32_t o = (uint32_t)(x0 - 4) & 3u;
  const uint8_t* base = img + (long long)(y0 - 4) * step + xb;
  constexpr int kWG = 0x000100FF;                                 // weights (-1, 0, +1, 0) on bytes 0..3
  constexpr int kWH = 0x00010201;                                 // weights ( 1, 2,  1, 0)
  int Gm2[7], Gm1[7], Hm2[7], Hm1[7];
  int a = 0, b = 0, c = 0;
#pragma unroll
  for (int t = 0; t < 9; t++) {
    const uint32_t* rp = reinterpret_cast<const uint32_t*>(base + (long long)t * step);
    const uint32_t d0 = rp[0] ^ 0x80808080u, d1 = rp[1] ^ 0x80808080u, d2 = rp[2] ^ 0x80808080u;
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, o);   // px -4 .. -1 (relative to x0)
    const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, o);   // px  0 ..  3
    const uint32_t w2 = d2 >> (8u * o);                          // px  4 in byte 0
    // windows starting at px j-1 for j = -3 .. 3 (the 4th byte has weight 0)
    const uint32_t win[7] = {w0, __builtin_amdgcn_alignbyte(w1, w0, 1), __builtin_amdgcn_alignbyte(w1, w0, 2),
                             __builtin_amdgcn_alignbyte(w1, w0, 3), w1, __builtin_amdgcn_alignbyte(w2, w1, 1),
                             __builtin_amdgcn_alignbyte(w2, w1, 2)};
    int G[7], H[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      G[j] = __builtin_amdgcn_sdot4((int)win[j], kWG, 0, false);
      H[j] = __builtin_amdgcn_sdot4((int)win[j], kWH, 0, false);
    }
    if (t >= 2) {
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const int Ix = Gm2[j] + 2 * Gm1[j] + G[j];
        const int Iy = H[j] - Hm2[j];
        a += Ix * Ix;
        b += Iy * Iy;
        c += Ix * Iy;
      }
    }
#pragma unroll
    for (int j = 0; j < 7; j++) { Gm2[j] = Gm1[j]; Gm1[j] = G[j]; Hm2[j] = Hm1[j]; Hm1[j] = H[j]; }
  }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

__global__ __launch_bounds__(256) void k_thr_harris(OrbGeometry g, FrameSrc src, const uint8_t* pyr,
                                                    const uint32_t* cand_cnt, const uint32_t* cand_key,
                                                    const uint8_t* cand_sc, const uint2* cmap,
                                                    uint32_t* s1_cnt, uint4* s1, uint32_t* status, int do_harris) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t thr_s, lcount;
  // Launched with 256 threads when the candidate lists are the dense ones (tens of thousands per level) and with one
  // wave when they come from the output-sensitive FAST pass (about a thousand): the stage is then a chain of short,
  // latency-bound phases, and four times as many (frame, level)s in flight beat four waves idling at the barriers.
  const int l = blockIdx.x, fi = blockIdx.y, slot = src.slot0 + fi, tid = threadIdx.x;
  const uint32_t nt = blockDim.x;
  if (l >= g.nlevels) return;
  const OrbLevelInfo L = g.lv[l];
  uint32_t n = cand_cnt[slot * kOrbLevels + l];
  const uint2 cm = cmap[slot * kOrbLevels + l];
  if (n > cm.y) {
    if (tid == 0) atomicOr(&status[slot], kStatusOverflow);
    n = cm.y;
  }
  const uint32_t* keys = cand_key + cm.x;
  const uint8_t* scs = cand_sc + cm.x;   // 16-byte aligned (every region starts at a multiple of 16 entries)
  uint4* out = s1 + (long long)slot * g.s1_total + L.s1_off;
  for (uint32_t b = tid; b < 256u; b += nt) hist[b] = 0;
  if (tid == 0) lcount = 0;
  __syncthreads();
  // 16 scores per lane and load; bytes past n inside the last group are stale and masked by index
  for (uint32_t i = 16u * tid; i < n; i += 16u * nt) {
    const uint4 v = *reinterpret_cast<const uint4*>(scs + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int b = 0; b < 16; b++)
      if (i + b < n) atomicAdd(&hist[(w[b >> 2] >> (8 * (b & 3))) & 255u], 1u);
  }
  __syncthreads();
  if (tid < 64) {
    // KeyPointsFilter::retainBest(keypoints, 2 * featuresNum): keep all >= the (2N)-th largest score = the largest
    // score b with (number of candidates scoring >= b) >= 2N.  One wave: lane i owns bins 4i .. 4i+3.
    const uint32_t keep = 2u * (uint32_t)L.quota;
    const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
    uint32_t above = h0 + h1 + h2 + h3;              // -> candidates in the bins of lanes > tid
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_down(above, o);
      if (tid + o < 64) above += up;
    }
    above -= h0 + h1 + h2 + h3;
    const uint32_t c3 = above + h3, c2 = c3 + h2, c1 = c2 + h1, c0 = c1 + h0;   // candidates scoring >= 4i+3 .. 4i
    const int mine = c3 >= keep ? 4 * tid + 3 : c2 >= keep ? 4 * tid + 2 : c1 >= keep ? 4 * tid + 1 : c0 >= keep ? 4 * tid : -1;
    const unsigned long long ok = __ballot(mine >= 0);
    uint32_t thr = 0;                                  // n <= 2N: keep everything
    if (n > keep) {
      thr = 256;                                       // keep == 0: drop all
      if (keep > 0 && ok) thr = (uint32_t)__shfl(mine, 63 - __builtin_clzll(ok));
    }
    if (tid == 0) thr_s = thr;
  }
  __syncthreads();
  const uint32_t thr = thr_s;
  int pitch;
  const uint8_t* img = level_ptr(g, src, pyr, fi, l, &pitch);
  // pass 1: compact the kept candidates (score >= thr) into the stage-1 list, response pending.  Kept candidates are
  // a few hundred out of tens of thousands: most waves see none and move on.
  for (uint32_t i0 = 0; i0 < n; i0 += 16u * nt) {
    const uint32_t i = i0 + 16u * tid;
    uint32_t mask = 0;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) {
      v = *reinterpret_cast<const uint4*>(scs + i);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int b = 0; b < 16; b++)
        if (i + b < n && ((w[b >> 2] >> (8 * (b & 3))) & 255u) >= thr) mask |= 1u << b;
    }
    if (__ballot(mask != 0u) == 0ull) continue;
    uint32_t k = reserve_packed(__popc(mask), &lcount, tid & 63);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    while (mask) {
      const int b = __ffs(mask) - 1;
      mask &= mask - 1;
      if (k < (uint32_t)kS1Cap)
        out[k] = make_uint4(keys[i + b], 0u, (w[b >> 2] >> (8 * (b & 3))) & 255u, 0u);
      k++;
    }
  }
  __syncthreads();
  // pass 2: Harris response on dense lanes
  // (batches leave this pass to k_harris_flat: there the kept candidates of ALL (frame, level)s are spread evenly over
  // waves of one round each, instead of one wave per (frame, level) walking up to four dependent rounds)
  const uint32_t kept = !do_harris ? 0u : min(lcount, (uint32_t)kS1Cap);
  for (uint32_t i = tid; i < kept; i += nt) {
    const uint32_t key = out[i].x;
    const float r = harris_at(img, pitch, key & 0xFFFF, key >> 16);
    out[i].y = __float_as_uint(r);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t m = lcount;
    if (m > (uint32_t)kS1Cap) { atomicOr(&status[slot], kStatusOverflow); m = kS1Cap; }
    s1_cnt[slot * kOrbLevels + l] = m;
  }
}

// HarrisResponses of the stage-1 lists, balanced: the lists of a frame are cut into units of 64 entries, level l getting
// hw.n[l] units (enough for 2N x 1.25 entries: retainBest(2N) keeps 2N plus the ties of the last score; a longer list is
// walked in steps of hw.n[l] x 64), one wave per unit.  k_thr_harris with one wave per (frame, level) spent four fifths
// of its time here -- level 0 alone is four dependent rounds of 27 scattered loads at 4 waves per SIMD -- 0.39 ms of the
// 0.56 ms selection stage per 2048 720p frames.
struct HarrisUnits {
  int base[kOrbLevels + 1];    // first unit of level l; base[nlevels] = units per frame
  int n[kOrbLevels];
};
__global__ __launch_bounds__(256) void k_harris_flat(OrbGeometry g, FrameSrc src, const uint8_t* pyr, HarrisUnits hw,
                                                     const uint32_t* __restrict__ s1_cnt, uint4* s1) {
  const int fi = blockIdx.y, slot = src.slot0 + fi, lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= hw.base[g.nlevels]) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < kOrbLevels; i++)
    if (i < g.nlevels && u >= hw.base[i]) l = i;
  const OrbLevelInfo L = g.lv[l];
  const uint32_t kept = min(s1_cnt[slot * kOrbLevels + l], (uint32_t)kS1Cap);
  uint4* out = s1 + (long long)slot * g.s1_total + L.s1_off;
  int pitch;
  const uint8_t* img = level_ptr(g, src, pyr, fi, l, &pitch);
  const uint32_t step = 64u * (uint32_t)hw.n[l];
  for (uint32_t i = 64u * (uint32_t)(u - hw.base[l]) + lane; i < kept; i += step) {
    const uint32_t key = out[i].x;
    out[i].y = __float_as_uint(harris_at(img, pitch, key & 0xFFFF, key >> 16));
  }
}

// ------------------------------------------------------------------ K7: retainBest(N) by Harris, canonical order
// One workgroup per frame walks the levels so the final list is ordered (level, y, x).
constexpr int kSelCap = kKpCap;
__global__ __launch_bounds__(256) void k_select(OrbGeometry g, int slot0, const uint32_t* s1_cnt, const uint4* s1,
                                                msf_keypoint* kp, uint32_t* kp_cnt, uint32_t* status) {
  __shared__ uint4 kept[kSelCap];
  // the two counting loops below read every response / key once per element: staged in LDS and read four at a time
  // (they were global loads: 63 us of the 0.45 ms single-pair call)
  constexpr int kRespCap = 4096;
  __shared__ __attribute__((aligned(16))) float resp[kRespCap];
  __shared__ __attribute__((aligned(16))) uint32_t keys[kSelCap];
  __shared__ uint32_t nkept;
  const int slot = slot0 + blockIdx.x, tid = threadIdx.x;
  msf_keypoint* out = kp + (long long)slot * kKpCap;
  uint32_t base = 0;
  bool overflow = false;
  for (int l = 0; l < g.nlevels; l++) {
    const OrbLevelInfo L = g.lv[l];
    const uint32_t n = s1_cnt[slot * kOrbLevels + l];
    const uint4* in = s1 + (long long)slot * g.s1_total + L.s1_off;
    const uint32_t keep = (uint32_t)L.quota;
    if (tid == 0) nkept = 0;
    const bool staged = n > keep && n <= (uint32_t)kRespCap;
    const uint32_t n4 = (n + 3) >> 2;
    if (staged)
      for (uint32_t i = tid; i < 4 * n4; i += 256) resp[i] = i < n ? __uint_as_float(in[i].y) : -INFINITY;
    __syncthreads();
    // retainBest(N): v is kept iff fewer than N entries are strictly greater than v
    for (uint32_t i = tid; i < n; i += 256) {
      const uint4 e = in[i];
      bool k = true;
      if (n > keep) {
        const float v = __uint_as_float(e.y);
        uint32_t greater = 0;
        if (staged) {
          const float4* r4 = reinterpret_cast<const float4*>(resp);
          for (uint32_t j = 0; j < n4; j++) {
            const float4 r = r4[j];
            greater += (uint32_t)(r.x > v) + (uint32_t)(r.y > v) + (uint32_t)(r.z > v) + (uint32_t)(r.w > v);
          }
        } else {
          for (uint32_t j = 0; j < n; j++) greater += (__uint_as_float(in[j].y) > v) ? 1u : 0u;
        }
        k = greater < keep;
      }
      if (k) {
        const uint32_t q = atomicAdd(&nkept, 1u);
        if (q < (uint32_t)kSelCap) kept[q] = e;
      }
    }
    __syncthreads();
    uint32_t m = nkept;
    if (m > (uint32_t)kSelCap) { overflow = true; m = kSelCap; }
    if (base + m > (uint32_t)kKpCap) { overflow = true; m = kKpCap - base; }
    // canonical order inside the level: rank by (y, x)
    const uint32_t m4 = (m + 3) >> 2;
    for (uint32_t i = tid; i < 4 * m4; i += 256) keys[i] = i < m ? kept[i].x : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += 256) {
      const uint4 e = kept[i];
      uint32_t rank = 0;
      const uint4* k4 = reinterpret_cast<const uint4*>(keys);
      for (uint32_t j = 0; j < m4; j++) {
        const uint4 kk = k4[j];
        rank += (uint32_t)(kk.x < e.x) + (uint32_t)(kk.y < e.x) + (uint32_t)(kk.z < e.x) + (uint32_t)(kk.w < e.x);
      }
      msf_keypoint k;
      k.lx = e.x & 0xFFFF;
      k.ly = e.x >> 16;
      k.x = (float)k.lx * L.scale;   // allKeypoints[i].pt *= scale
      k.y = (float)k.ly * L.scale;
      k.response = __uint_as_float(e.y);
      k.angle = -1.f;
      k.octave = l;
      k.fast_score = (int)e.z;
      out[base + rank] = k;
    }
    base += m;
    __syncthreads();
  }
  if (tid == 0) {
    kp_cnt[slot] = base;
    if (overflow) atomicOr(&status[slot], kStatusOverflow);
  }
}

// ------------------------------------------------------------------ K8+K9+K10: orientation, blur, steered rBRIEF
// cv::fastAtan2 scalar path (core mathfuncs_core.simd.hpp atan_f32)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float k = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
              p7 = -0.04432655554792128f * k;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// sin/cos of t in [0, ~2*pi]: Cody-Waite reduction by pi/2 + fdlibm kernel polynomials in IEEE
// double (plain mul/add, no FMA), rounded once to f32.  Same operation sequence as the CPU side.
__device__ __forceinline__ void det_sincosf(float t, float* s, float* c) {
  const double x = (double)t;
  const int k = (int)(x * 6.36619772367581382433e-01 + 0.5);
  const double kd = (double)k;
  const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
  const double z = r * r;
  double p = 1.58969099521155010221e-10;
  p = p * z + -2.50507602534068634195e-08;
  p = p * z + 2.75573137070700676789e-06;
  p = p * z + -1.98412698298579493134e-04;
  p = p * z + 8.33333333332248946124e-03;
  p = p * z + -1.66666666666666324348e-01;
  const double sr = r + (r * z) * p;
  double q = -1.13596475577881948265e-11;
  q = q * z + 2.08757232129817482790e-09;
  q = q * z + -2.75573143513906633035e-07;
  q = q * z + 2.48015872894767294178e-05;
  q = q * z + -1.38888888888741095749e-03;
  q = q * z + 4.16666666666666019037e-02;
  const double cr = (1.0 - 0.5 * z) + (z * z) * q;
  double sv, cv;
  switch (k & 3) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
  }
  *s = (float)sv;
  *c = (float)cv;
}

constexpr int PR = 22;             // patch radius: 19 (rotated pattern reach) + 3 (blur taps)
constexpr int PD = 2 * PR + 1;     // 45
constexpr int PP = 48;             // raw patch pitch
constexpr int HP = 40;             // columns of the row-blurred patch
constexpr int HC = 46;             // its column pitch in u16: 45 rows + 1; 92 B puts the ten column groups of a blur task on
                                   // (nearly) distinct LDS banks (48 would put them all on one)

// HALF_UP: the blur's column pass rounds half up (MSF_FLAG_BLUR_TIE_HALF_UP, and always with SUM256) instead of half to
// even; SUM256: OpenCV's fixed-point kernel 18 34 48 56 48 34 18 instead of 18 34 49 55 49 34 18.  Compile-time: the
// rounding is three instructions without a branch and the taps are literals.
template <bool HALF_UP, bool SUM256>
__global__ __launch_bounds__(256) void k_describe(OrbGeometry g, FrameSrc src, const uint8_t* pyr,
                                                  msf_keypoint* kp, const uint32_t* kp_cnt, uint8_t* desc) {
  __shared__ __attribute__((aligned(16))) uint8_t raw_s[4][PD * PP + 16];
  // row-blurred patch, TRANSPOSED: column c (= sample x + 19) holds its 45 rows as consecutive u16 (pitch HC, even), so the
  // seven vertical taps of a sample are four consecutive dwords (two LDS reads instead of seven)
  __shared__ __attribute__((aligned(16))) uint16_t hb_s[4][HP * HC];
  __shared__ uint32_t disc_s[2 * kDiscTasks];
  // the rBRIEF table sits in LDS: read from constant memory per key point it is a VECTOR load behind the next patch's
  // prefetch, and waiting for it (vmcnt counts in order) waited for the prefetch too -- every key point then paid a full
  // memory latency (6 us per key point and wave; the kernel was bound by neither its bytes nor its instructions)
  __shared__ __attribute__((aligned(16))) int pat_s[256];
  const int fi = blockIdx.y, slot = src.slot0 + fi;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (scalar: key-point indices and record addresses are then scalar too)
  for (int i = threadIdx.x; i < 2 * kDiscTasks; i += 256) disc_s[i] = c_disc[i];
  pat_s[threadIdx.x] = reinterpret_cast<const int*>(c_pattern)[threadIdx.x];
  // base pointer and pitch of this frame's levels: the level of a key point is a vector register to the compiler, and the
  // geometry struct indexed by it was two more vector loads, waited for in full, in front of every patch prefetch
  __shared__ unsigned long long lvl_base_s[kOrbLevels];
  __shared__ int lvl_pitch_s[kOrbLevels];
  if (threadIdx.x < kOrbLevels) {
    int pt = 0;
    const uint8_t* bp = level_ptr(g, src, pyr, fi, threadIdx.x < g.nlevels ? (int)threadIdx.x : 0, &pt);
    lvl_base_s[threadIdx.x] = (unsigned long long)bp;
    lvl_pitch_s[threadIdx.x] = pt;
  }
  __syncthreads();
  const uint32_t count = min(kp_cnt[slot], (uint32_t)kKpCap);
  uint8_t* raw = raw_s[wave];
  uint16_t* hb = hb_s[wave];
  uint32_t* raw32 = reinterpret_cast<uint32_t*>(raw);
  // Each wave owns its LDS patch and walks its own key points: waves never wait for each other.  LDS operations of
  // one wave are executed in issue order, so a wavefront-scope fence (compiler ordering) is all the staging needs.
  // Fetching the 45 x 48-byte patch is 63 % of this kernel when it is waited for (the patches of a batch amount to one
  // more read of the whole pyramid, at HBM speed): the patch of the wave's NEXT key point is requested into registers
  // before the current one is processed, and its position one key point earlier still.
  constexpr int NPL = (PD * (PP / 4) + 63) / 64;          // dword loads per lane and patch
  const uint32_t kstride = gridDim.x * 4;
  uint32_t k = blockIdx.x * 4 + wave;
  uint32_t pre[NPL];
  int xo_pre = 0;
  // position (level, x, y) of key point k_ -> packed; patch loads of a packed position -> pre[], xo_pre
  auto meta = [&](uint32_t k_) -> uint3 {
    if (k_ >= count) return make_uint3(0u, 0u, 0u);
    const msf_keypoint* Kp = kp + (long long)slot * kKpCap + k_;
    return make_uint3((uint32_t)Kp->octave, (uint32_t)Kp->lx, (uint32_t)Kp->ly);
  };
  // A wave walks ONE key point at a time: level and position are the same in every lane (they come from one address), and
  // as scalars the patch is  s[base] + 32-bit lane offset  per load -- row r = i / 12 by a multiply-shift (exact for
  // i < 576: 43691 = ceil(2^19 / 12)), offset r (pitch - 48) + 4 i.  (r04 formed r, the column and a 64-bit address per
  // lane and load: 105 of the ~510 vector instructions per key point of a kernel that is 0.82 VALU-busy.)
  auto issue = [&](uint3 m) {
    const uint32_t lv = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.x);
    const int mx = __builtin_amdgcn_readfirstlane((int)m.y), my = __builtin_amdgcn_readfirstlane((int)m.z);
    const uint32_t pitch_ = (uint32_t)__builtin_amdgcn_readfirstlane(lvl_pitch_s[lv]);
    const unsigned long long b64 = lvl_base_s[lv];
    const unsigned long long img_ = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
    typedef const __attribute__((address_space(1))) uint8_t* gptr_t;   // global, not generic: a flat load counts as an LDS op too
    // raw patch, radius 22, fetched as aligned dwords: columns ax .. ax+47 hold x = cx-22 .. cx+22 at byte
    // offset xo (keypoints sit >= 31 px inside the level, so this never leaves the row)
    const int ax = (mx - PR) & ~3;
    xo_pre = (mx - PR) - ax;
    gptr_t base = (gptr_t)(img_ + (unsigned long long)((long long)(my - PR) * (long long)pitch_ + ax));
    const uint32_t pm = pitch_ - (uint32_t)PP;
#pragma unroll
    for (int u = 0; u < NPL; u++) {
      const uint32_t i = (uint32_t)lane + 64u * (uint32_t)u;
      const uint32_t r = (i * 43691u) >> 19;
      const uint32_t off = mad_u24(r, pm, 4u * i);
      pre[u] = i < (uint32_t)(PD * (PP / 4)) ? *(const __attribute__((address_space(1))) uint32_t*)(base + off) : 0u;
    }
  };
  // row blur: lane -> (row pair rl within a block of six, column group gq); dword offsets of its reads and writes
  uint32_t* hb32 = reinterpret_cast<uint32_t*>(hb);
  const int blur_rl = (lane * 26) >> 8, blur_gq = lane - 10 * blur_rl;       // lane / 10, lane % 10 (lanes 60..63: 6, 0..3)
  const int blur_rd = blur_rl * (2 * (PP / 4)) + blur_gq, blur_wr = blur_gq * (4 * (HC / 2)) + blur_rl;
  const bool blur_last = 18 + blur_rl < 23;                                    // pass 3 holds row pairs 18 .. 24
  // descriptor samples: LDS byte address of hb[19][19 - 3], minus what the magic-constant sums carry (see there)
  const uint32_t samp_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)hb +
                             2u * (uint32_t)((PR - 3) * HC + (PR - 3)) - 2u * HC * 0x400000u - 2u * 0x4B400000u;
  uint3 m_cur = meta(k), m_next = meta(k + kstride);
  if (k < count) issue(m_cur);
  for (; k < count; k += kstride) {
    const bool active = true;
    msf_keypoint* K = kp + (long long)slot * kKpCap + k;
    const int l = (int)m_cur.x, cx = (int)m_cur.y, cy = (int)m_cur.z;
    (void)l; (void)cx; (void)cy;
    const int xo = xo_pre;
#pragma unroll
    for (int u = 0; u < NPL; u++) {
      const int i = lane + 64 * u;
      if (i < PD * (PP / 4)) raw32[i] = pre[u];
    }
    m_cur = m_next;
    if (k + kstride < count) issue(m_cur);          // next patch: in flight while this key point is processed
    m_next = meta(k + 2 * kstride);
    MSF_WAVE_SYNC();
    float angle = 0.f;
    if (active) {
      // ICAngles (orb.cpp): m10 = sum u * p, m01 = sum v * p over the 31-px disc, integer, so the summation order is
      // free.  A task is one dword of 4 pixels of one disc row: m10 += dot4(p - 128, u weights) (the -128 cancels:
      // the u weights of a whole row sum to 0) and m01 += v * dot4(p, membership).  4 passes of the wave.
      int m10 = 0, m01 = 0;
      {
        // pixel (u, v) sits at byte xo + 22 + u of raw row 22 + v; the windows start at u = -16 + 4k
        const int dw0 = 1 + ((xo + 2) >> 2);
        const uint32_t shf = (uint32_t)(xo + 2) & 3u;
#pragma unroll
        for (int it = 0; it < kDiscTasks / 64; it++) {
          const int t = it * 64 + lane;
          const int row = t >> 3, k = t & 7;              // row = v + 15 (rows 31 are padding: zero weights); row + 7 <= 38
          const uint32_t* d = raw32 + (row + 7) * (PP / 4) + dw0 + k;
          const uint32_t pxw = __builtin_amdgcn_alignbyte(d[1], d[0], shf) ^ 0x80808080u;
          const uint32_t wu = disc_s[2 * t], wv = disc_s[2 * t + 1];
          // both moments as signed dot products of p - 128 with the byte weights u and v (0 outside the disc): the -128
          // cancels over a row for m10 (the u of a row sum to 0) and over the disc for m01 (rows v and -v are equally long)
          m10 = __builtin_amdgcn_sdot4((int)pxw, (int)wu, m10, false);
          m01 = __builtin_amdgcn_sdot4((int)pxw, (int)wv, m01, false);
        }
      }
      // wave sums by DPP (the shuffle form is twelve LDS round trips through ds_bpermute)
      m10 = (int)__builtin_amdgcn_readlane((int)wave_incl_scan((uint32_t)m10), 63);
      m01 = (int)__builtin_amdgcn_readlane((int)wave_incl_scan((uint32_t)m01), 63);
      angle = fast_atan2_deg((float)m01, (float)m10);
      // 7-tap row pass of GaussianBlur(7x7, sigma 2) in its 8u integer form (18 34 49 55 49 34 18, or OpenCV's bit-exact
      // fixed-point kernel 18 34 48 56 48 34 18).  A task = TWO rows x 4 output columns from 2 x 4 aligned dwords: output j
      // is the taps laid over bytes j .. j + 6 of the 12-byte window, i.e. v_dot4_u32_u8 of each window dword with the
      // taps SHIFTED into place (10 dot products per 4 outputs, no per-output byte alignment).  Lane -> (row pair within a
      // block of six, column group), so every LDS address of the four passes is lane base + immediate.
      {
        constexpr uint32_t T0 = 18, T1 = 34, T2 = SUM256 ? 48 : 49, T3 = SUM256 ? 56 : 55;      // taps T0 T1 T2 T3 T2 T1 T0
        constexpr uint32_t A0 = T0 | T1 << 8 | T2 << 16 | T3 << 24, B0 = T2 | T1 << 8 | T0 << 16;
        constexpr uint32_t A1 = T0 << 8 | T1 << 16 | T2 << 24, B1 = T3 | T2 << 8 | T1 << 16 | T0 << 24;
        constexpr uint32_t A2 = T0 << 16 | T1 << 24, B2 = T2 | T3 << 8 | T2 << 16 | T1 << 24, C2 = T0;
        constexpr uint32_t A3 = T0 << 24, B3 = T1 | T2 << 8 | T3 << 16 | T2 << 24, C3 = T1 | T0 << 8;
        auto row4 = [&](const uint32_t* d, uint32_t* o) {
          const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
          const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, xo), e1 = __builtin_amdgcn_alignbyte(d2, d1, xo);
          const uint32_t e2 = __builtin_amdgcn_alignbyte(d3, d2, xo);
          o[0] = __builtin_amdgcn_udot4(e0, A0, __builtin_amdgcn_udot4(e1, B0, 0u, false), false);
          o[1] = __builtin_amdgcn_udot4(e0, A1, __builtin_amdgcn_udot4(e1, B1, 0u, false), false);
          o[2] = __builtin_amdgcn_udot4(e0, A2, __builtin_amdgcn_udot4(e1, B2, __builtin_amdgcn_udot4(e2, C2, 0u, false), false), false);
          o[3] = __builtin_amdgcn_udot4(e0, A3, __builtin_amdgcn_udot4(e1, B3, __builtin_amdgcn_udot4(e2, C3, 0u, false), false), false);
        };
        // 23 row pairs (rows 0 .. 45; row 45 is the pad row of hb and reads the 16 spare bytes behind the raw patch and
        // whatever follows them -- never sampled) x 10 column groups = 4 passes of 60 lanes; lanes 60 .. 63 repeat what
        // lanes 0 .. 3 do in the next pass (same values to the same place) except in the last pass, where lanes with
        // row pair >= 23 must not write
#pragma unroll
        for (int it = 0; it < 4; it++) {
          const uint32_t* d = raw32 + blur_rd + it * (12 * (PP / 4));
          uint32_t ou[4], ol[4];
          row4(d, ou);
          row4(d + PP / 4, ol);
          if (it < 3 || blur_last) {
            uint32_t* w = hb32 + blur_wr + it * 6;
#pragma unroll
            for (int j = 0; j < 4; j++) w[j * (HC / 2)] = ou[j] | (ol[j] << 16);     // each <= 255 * 257 = 65535
          }
        }
      }
    }
    MSF_WAVE_SYNC();
    if (active) {
      float a, b;
      float rad = angle;
      rad *= (float)(3.14159265358979323846 / 180.f);
      det_sincosf(rad, &b, &a);  // a = cos, b = sin
      // computeOrbDescriptors (orb.cpp), WTA_K = 2: lane handles tests 4*lane .. 4*lane+3
      const int4 pk = *reinterpret_cast<const int4*>(&pat_s[lane * 4]);
      const int words[4] = {pk.x, pk.y, pk.z, pk.w};
      uint32_t nib = 0;
      constexpr uint32_t K2 = SUM256 ? 48 : 49, K3 = SUM256 ? 56 : 55;
      constexpr uint32_t P0 = 18u | 34u << 16, P1 = K2 | K3 << 16, P2 = K2 | 34u << 16, P3 = 18u;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        uint32_t val[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float px = (float)(signed char)(words[t] >> (16 * e));
          const float py = (float)(signed char)(words[t] >> (16 * e + 8));
          // cvRound by the magic constant: for |v| < 2^22, v + 1.5 * 2^23 is v rounded to the nearest integer, ties to even,
          // in the low mantissa bits -- the sum's bit pattern is 0x4B400000 + round(v)
          const uint32_t tx = __float_as_uint((px * a - py * b) + 12582912.f);
          const uint32_t ty = __float_as_uint((px * b + py * a) + 12582912.f);
          // byte address of hb[column 19 + ix][row 19 + iy - 3]: 2 * HC * ix + 2 * iy + constant; v_mad_u32_u24 takes the low
          // 24 bits of tx (0x400000 + ix) and everything else is folded into samp_base (arithmetic modulo 2^32)
          const uint32_t addr = mad_u24(tx, 2u * HC, (ty << 1) + samp_base);
          const __attribute__((address_space(3))) uint32_t* q =
              (const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(addr & ~3u);   // (uintptr_t: the host pass sees a 64-bit pointer)
          const uint32_t sh = addr & 2u;                 // the seven u16 start in the upper half of the first dword
          const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
          const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
          const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, sh), w3 = __builtin_amdgcn_alignbyte(0u, d3, sh);
          const uint32_t sum = udot2_u16(w0, P0, udot2_u16(w1, P1, udot2_u16(w2, P2, udot2_u16(w3, P3, 0u))));
          // (sum + 32768) >> 16, ties up or to even
          const uint32_t r = HALF_UP ? (sum + 32768u) >> 16 : (sum + 0x7FFFu + ((sum >> 16) & 1u)) >> 16;
          val[e] = min(r, 255u);
        }
        nib |= (uint32_t)(val[0] < val[1]) << t;
      }
      const uint32_t hi = __shfl_down(nib, 1);
      if ((lane & 1) == 0) desc[((long long)slot * kKpCap + k) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
      if (lane == 0) K->angle = angle;
    }
    MSF_WAVE_SYNC();
  }
}

// ------------------------------------------------------------------ K11: brute-force Hamming 2-NN + ratio + ordered compaction
constexpr int kTrainChunk = 1024;  // train descriptors staged per LDS pass (32 KB)
__global__ __launch_bounds__(256) void k_match(int n_pairs, const int32_t* slot_a, const int32_t* slot_b,
                                               const msf_keypoint* kp, const uint32_t* kp_cnt, const uint8_t* desc,
                                               const uint32_t* status, float ratio, msf_match* out, int cap,
                                               int32_t* n_out, int chunk, int slot_base, int max_slots) {
  __shared__ __attribute__((aligned(16))) unsigned long long train[kTrainChunk * 4];
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t running;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // slot arrays come from the caller's device memory and cannot be checked on the host: a slot outside the handle's
  // range gives n_out = -1 for that pair instead of an out-of-bounds read
  const int sa = slot_a ? slot_a[pair] : slot_base + pair;
  const int sb = slot_b ? slot_b[pair] : slot_base + n_pairs + pair;
  if (sa < 0 || sa >= max_slots || sb < 0 || sb >= max_slots || ((status[sa] | status[sb]) & kStatusOverflow)) {
    if (tid == 0) n_out[pair] = -1;
    return;
  }
  const uint32_t n1 = min(kp_cnt[sa], (uint32_t)kKpCap), n2 = min(kp_cnt[sb], (uint32_t)kKpCap);
  const unsigned long long* tsrc = reinterpret_cast<const unsigned long long*>(desc + (long long)sb * kKpCap * 32);
  if (tid == 0) running = 0;
  __syncthreads();
  const msf_keypoint* ka = kp + (long long)sa * kKpCap;
  const msf_keypoint* kb = kp + (long long)sb * kKpCap;
  msf_match* o = out + (long long)pair * cap;
  for (uint32_t q0 = 0; q0 < n1; q0 += 256) {
    const uint32_t q = q0 + tid;
    const bool have_q = q < n1;
    unsigned long long q0w = 0, q1w = 0, q2w = 0, q3w = 0;
    if (have_q) {
      const unsigned long long* qd =
          reinterpret_cast<const unsigned long long*>(desc + ((long long)sa * kKpCap + q) * 32);
      q0w = qd[0]; q1w = qd[1]; q2w = qd[2]; q3w = qd[3];
    }
    int d0 = 0x7fffffff, d1 = 0x7fffffff, i0 = -1;
    for (uint32_t t0 = 0; t0 < n2; t0 += chunk) {
      const uint32_t tn = min(n2 - t0, (uint32_t)chunk);
      __syncthreads();
      for (uint32_t i = tid; i < tn * 4; i += 256) train[i] = tsrc[(size_t)t0 * 4 + i];
      __syncthreads();
      if (have_q) {
        for (uint32_t t = 0; t < tn; t++) {
          const unsigned long long* tr = &train[t * 4];
          const int d = __popcll(q0w ^ tr[0]) + __popcll(q1w ^ tr[1]) + __popcll(q2w ^ tr[2]) + __popcll(q3w ^ tr[3]);
          // batchDistance K=2 insertion: strict '<' keeps the lower train index on ties
          if (d < d1) {
            if (d < d0) { d1 = d0; d0 = d; i0 = (int)(t0 + t); }
            else d1 = d;
          }
        }
      }
    }
    // train sets of < 2 give no matches (reference: UB at matches[i][1]); featurematcher.cpp:32
    const bool pass = have_q && n2 >= 2 && ((float)d0 < ratio * (float)d1);
    // ordered compaction (query order)
    const unsigned long long bal = __ballot(pass);
    const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    uint32_t off = running;
    for (int w = 0; w < wave; w++) off += wave_cnt[w];
    if (pass) {
      const uint32_t idx = off + before;
      if (idx < (uint32_t)cap) {
        msf_match m;
        m.x1 = (int)ka[q].x; m.y1 = (int)ka[q].y;      // static_cast<int>(kp.pt.x) (:33-38)
        m.x2 = (int)kb[i0].x; m.y2 = (int)kb[i0].y;
        o[idx] = m;
      }
    }
    __syncthreads();
    if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) n_out[pair] = (int32_t)running;
}

// Small batches (the drop-in call is ONE pair): the same matching spread over kSplitBlocks workgroups per pair.  A
// workgroup takes 64 queries; its four waves scan a quarter of the train descriptors each and the partial 2-NN results
// are merged by value with ties to the lower train index -- exactly what the sequential insertion above keeps.  The
// workgroup that finishes last (ticket counter) does the ordered compaction of the per-query results.
constexpr int kSplitBlocks = kKpCap / 64;
constexpr int kSplitMaxPairs = 128;
__global__ __launch_bounds__(256) void k_match_split(int n_pairs, const int32_t* slot_a, const int32_t* slot_b,
                                                     const msf_keypoint* kp, const uint32_t* kp_cnt,
                                                     const uint8_t* desc, const uint32_t* status, float ratio,
                                                     msf_match* out, int cap, int32_t* n_out, int chunk,
                                                     uint32_t* qres, uint32_t* done, int slot_base, int max_slots) {
  __shared__ __attribute__((aligned(16))) unsigned long long train[kTrainChunk * 4];
  __shared__ int part[3][4][64];
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t running, last_s;
  const int pair = blockIdx.y, qb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sa = slot_a ? slot_a[pair] : slot_base + pair;
  const int sb = slot_b ? slot_b[pair] : slot_base + n_pairs + pair;
  // uniform over the pair's workgroups: nobody takes a ticket (bad slots: see k_match)
  if (sa < 0 || sa >= max_slots || sb < 0 || sb >= max_slots || ((status[sa] | status[sb]) & kStatusOverflow)) {
    if (qb == 0 && tid == 0) n_out[pair] = -1;
    return;
  }
  const uint32_t n1 = min(kp_cnt[sa], (uint32_t)kKpCap), n2 = min(kp_cnt[sb], (uint32_t)kKpCap);
  const unsigned long long* tsrc = reinterpret_cast<const unsigned long long*>(desc + (long long)sb * kKpCap * 32);
  uint32_t* qr = qres + (long long)pair * kKpCap;
  const uint32_t q = (uint32_t)qb * 64u + lane;
  if ((uint32_t)qb * 64u < n1) {   // uniform
    const bool have_q = q < n1;
    unsigned long long q0w = 0, q1w = 0, q2w = 0, q3w = 0;
    if (have_q) {
      const unsigned long long* qd = reinterpret_cast<const unsigned long long*>(desc + ((long long)sa * kKpCap + q) * 32);
      q0w = qd[0]; q1w = qd[1]; q2w = qd[2]; q3w = qd[3];
    }
    int d0 = 0x7fffffff, d1 = 0x7fffffff, i0 = 0x7fffffff;
    for (uint32_t t0 = 0; t0 < n2; t0 += chunk) {
      const uint32_t tn = min(n2 - t0, (uint32_t)chunk);
      __syncthreads();
      for (uint32_t i = tid; i < tn * 4; i += 256) train[i] = tsrc[(size_t)t0 * 4 + i];
      __syncthreads();
      const uint32_t ta = (tn * wave) >> 2, tb = (tn * (wave + 1)) >> 2;   // this wave's quarter of the chunk
      for (uint32_t t = ta; t < tb; t++) {
        const unsigned long long* tr = &train[t * 4];
        const int d = __popcll(q0w ^ tr[0]) + __popcll(q1w ^ tr[1]) + __popcll(q2w ^ tr[2]) + __popcll(q3w ^ tr[3]);
        if (d < d1) {
          if (d < d0) { d1 = d0; d0 = d; i0 = (int)(t0 + t); }
          else d1 = d;
        }
      }
    }
    part[0][wave][lane] = d0;
    part[1][wave][lane] = i0;
    part[2][wave][lane] = d1;
    __syncthreads();
    if (wave == 0) {
      // quarters are visited in increasing train order inside a chunk but chunks interleave them: merge by
      // (distance, index), which is what one sequential pass keeps
      int b0 = 0x7fffffff, bi = 0x7fffffff, b1 = 0x7fffffff;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const int e0 = part[0][w][lane], ei = part[1][w][lane], e1 = part[2][w][lane];
        if (e0 < b0 || (e0 == b0 && ei < bi)) { b1 = min(b0, e1); b0 = e0; bi = ei; }
        else b1 = min(b1, e0);
        b1 = min(b1, e1);
      }
      const bool pass = have_q && n2 >= 2 && ((float)b0 < ratio * (float)b1);
      if (have_q) qr[q] = pass ? (0x80000000u | (uint32_t)bi) : 0u;
    }
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) last_s = atomicAdd(&done[pair], 1u) == (uint32_t)gridDim.x - 1u;
  __syncthreads();
  if (!last_s) return;
  __threadfence();
  // ordered compaction (query order) by the last workgroup of the pair
  if (tid == 0) { running = 0; done[pair] = 0; }
  __syncthreads();
  const msf_keypoint* ka = kp + (long long)sa * kKpCap;
  const msf_keypoint* kb = kp + (long long)sb * kKpCap;
  msf_match* o = out + (long long)pair * cap;
  for (uint32_t q0 = 0; q0 < n1; q0 += 256) {
    const uint32_t qq = q0 + tid;
    const uint32_t r = qq < n1 ? __hip_atomic_load(&qr[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const bool pass = (r & 0x80000000u) != 0u;
    const unsigned long long bal = __ballot(pass);
    const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    uint32_t off = running;
    for (int w = 0; w < wave; w++) off += wave_cnt[w];
    if (pass) {
      const uint32_t idx = off + before;
      if (idx < (uint32_t)cap) {
        const uint32_t t = r & 0x7fffffffu;
        msf_match m;
        m.x1 = (int)ka[qq].x; m.y1 = (int)ka[qq].y;      // static_cast<int>(kp.pt.x) (featurematcher.cpp:33-38)
        m.x2 = (int)kb[t].x; m.y2 = (int)kb[t].y;
        o[idx] = m;
      }
    }
    __syncthreads();
    if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) n_out[pair] = (int32_t)running;
}

// ================================================================== host side
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

// interpolationLinear<uchar>::getCoeffs (imgproc resize.cpp, INTER_LINEAR_EXACT)
static void make_table(int src, int dst, uint32_t* tab) {
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dst; d++) {
    const double f = scale * ((double)d + 0.5) - 0.5;
    const int i = (int)floor(f);
    if (i >= 0 && src > 1) {
      if (i < src - 1) {
        tab[d] = (uint32_t)i | ((uint32_t)cv_round_d((f - (double)i) * 256.0) << 16);
      } else {
        tab[d] = (uint32_t)(src - 1);   // replicate the last pixel: weight 0 on the (absent) right/bottom tap
      }
    } else {
      tab[d] = 0;                       // replicate the first pixel
    }
  }
}

OrbPipeline::~OrbPipeline() { destroy(); }

void OrbPipeline::destroy() {
  hipFree(d_tau_); hipFree(d_redo_); hipFree(d_qstat_); hipFree(d_walk_abort_);
  if (h_walk_abort_) hipHostFree(h_walk_abort_);
  d_tau_ = nullptr; d_redo_ = nullptr; d_qstat_ = nullptr; d_walk_abort_ = nullptr; h_walk_abort_ = nullptr;
  hipFree(d_pyr_); hipFree(d_tab_); hipFree(d_cand_cnt_); hipFree(d_cand_); hipFree(d_cand_sc_); hipFree(d_cmap_); hipFree(d_pool_cnt_); hipFree(d_qres_); hipFree(d_done_); hipFree(d_s1_cnt_); hipFree(d_s1_);
  hipFree(d_kp_); hipFree(d_desc_); hipFree(d_kp_cnt_); hipFree(d_status_);
  d_pyr_ = nullptr; d_tab_ = nullptr; d_cand_cnt_ = nullptr; d_cand_ = nullptr; d_cand_sc_ = nullptr; d_cmap_ = nullptr; d_pool_cnt_ = nullptr; d_qres_ = nullptr; d_done_ = nullptr; d_s1_cnt_ = nullptr;
  d_s1_ = nullptr; d_kp_ = nullptr; d_desc_ = nullptr; d_kp_cnt_ = nullptr; d_status_ = nullptr;
  if (ev_ok_) {
    for (auto& set : evr_)
      for (auto& e : set.ev) if (e) hipEventDestroy(e);
  }
  ev_ok_ = false;
}

#define MSF_HIP_TRY(expr)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e_);       \
  } while (0)

std::string OrbPipeline::init(int width, int height, int max_slots, bool blur_half_up, bool profile, bool dense_fast,
                              bool level_size_mul_inv, int stream_min_frames, bool blur_sum256, int work_frames) {
  if (width < 64 || height < 64 || width > 8192 || height > 8192) return "arg: ORB image size must be in [64, 8192] x [64, 8192]";
  if (max_slots < 2) return "arg: ORB max_slots < 2";
  max_slots_ = max_slots;
  // frames one extraction may hold: the pyramid, the candidate and stage-1 lists and the walker's state exist once per
  // frame of a CALL (work rows), only key points and descriptors once per feature slot
  work_frames_ = work_frames <= 0 || work_frames > max_slots ? max_slots : work_frames < 2 ? 2 : work_frames;
  half_up_ = blur_half_up;
  blur_sum256_ = blur_sum256;
  profile_ = profile;
  // MSF_ORB_FAST_TAU forces the first-pass FAST threshold (tests: a value no level can reach sends every level through
  // the check + dense second pass); MSF_FLAG_FAST_DENSE = 20 = the plain dense detector
  force_tau_ = dense_fast ? kFastT : 0;
  stream_min_frames_ = stream_min_frames;
  if (const char* e = getenv("MSF_ORB_FAST_TAU")) {
    const int v = atoi(e) & ~1;
    if (v >= kFastT && v <= 254) force_tau_ = v;
  }
  // MSF_ORB_RESIZE_GENERIC=1: k_resize reads every pixel's taps from its own dword pair (the fallback for level
  // geometries whose column table fails the shared-pair check; tests compare the two)
  if (const char* e = getenv("MSF_ORB_RESIZE_GENERIC")) resize_generic_ = atoi(e) != 0;
  // MSF_ORB_FAST_ONE_PART=1: the streaming FAST pass over all strips at the first threshold (no prediction, no refinement)
  if (const char* e = getenv("MSF_ORB_FAST_ONE_PART")) fast_two_part_ = atoi(e) == 0;
  // MSF_ORB_UNFUSED=1: the pyramid by k_resize and one FAST-only walker launch over all levels (the fallback for level
  // geometries whose tables fail the host checks below; tests compare it with the fused default, in which the walker of
  // level l - 1 also makes level l)
  if (const char* e = getenv("MSF_ORB_UNFUSED")) fused_ = atoi(e) == 0;
  if (getenv("MSF_TEST_HOOKS"))
    if (const char* e = getenv("MSF_ORB_TEST_STALL_FRAME")) test_stall_frame_ = atoi(e);
  // MSF_ORB_WALK_PER_LEVEL=1: the fused walker as one launch per level instead of one launch over all levels (tests
  // compare the two: the in-launch dependency waits are then met the moment a unit starts)
  if (const char* e = getenv("MSF_ORB_WALK_PER_LEVEL")) walk_per_level_ = atoi(e) != 0;
  // MSF_ORB_TAU_PREDICT: percent of the needed corner density the prediction of a level's first threshold from the level
  // above keeps (0 = sample every level)
  if (const char* e = getenv("MSF_ORB_TAU_PREDICT")) {
    const int v = atoi(e);
    if (v >= 0 && v <= 2000) tau_predict_pct_ = v;
  }
  // MSF_ORB_TAU2_MARGIN_PCT: the second estimate's safety margin in percent of 2N (tests: a few percent makes it overshoot,
  // so that levels fail the check and take the dense second pass)
  tau2_margin_pct_ = kTau2MarginPct;
  if (const char* e = getenv("MSF_ORB_TAU2_MARGIN_PCT")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 10000) tau2_margin_pct_ = v;
  }
  OrbGeometry& g = g_;
  g.nlevels = kOrbLevels;
  g.w0 = width;
  g.h0 = height;
  // ORB_Impl::detectAndCompute: layer scales and sizes; computeKeyPoints: per-level quotas and umax
  const float scale_factor_f = 1.2f;
  const double scale_factor = (double)scale_factor_f;
  const int nfeatures = 500;
  {
    const float factor = (float)(1.0 / scale_factor);
    float nd = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)g.nlevels));
    int sum = 0;
    for (int l = 0; l < g.nlevels - 1; l++) {
      g.lv[l].quota = cv_round_f(nd);
      sum += g.lv[l].quota;
      nd *= factor;
    }
    g.lv[g.nlevels - 1].quota = nfeatures - sum > 0 ? nfeatures - sum : 0;
  }
  {
    const int half = 15;
    int umax[17] = {0};
    const int vmax = (int)floor(half * sqrt(2.f) / 2 + 1);
    const int vmin = (int)ceil(half * sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
    for (int v = 0; v < 16; v++) g.umax[v] = umax[v];
  }
  long long pix = 0;
  int cand = 0, prim = 0, tiles = 0, strips = 0, tab = 0, s1 = 0;
  g.max_level_tiles = 0;
  for (int l = 0; l < g.nlevels; l++) {
    OrbLevelInfo& L = g.lv[l];
    const float s = (float)pow(scale_factor, (double)l);
    const float inv = 1.0f / s;
    L.scale = s;
    // ORB_Impl::detectAndCompute (OpenCV 3.x / 4.x): Size sz(cvRound(image.cols/scale), cvRound(image.rows/scale));
    // MSF_FLAG_LEVEL_SIZE_MUL_INV selects the 2.4-era cvRound(cols * (1.f / scale)) instead (they differ for a few odd
    // sizes, e.g. width 69 -> 57 vs 58 at level 1; never for 640 / 480 / 1280 / 720 / 1920 / 1080)
    L.w = level_size_mul_inv ? cv_round_f((float)width * inv) : cv_round_f((float)width / s);
    L.h = level_size_mul_inv ? cv_round_f((float)height * inv) : cv_round_f((float)height / s);
    L.pitch = (L.w + 15) & ~15;
    L.pix_off = pix;
    if (l > 0) pix += (long long)L.pitch * L.h;
    // strict 3x3 maxima are at most w*h/4; w*h/8 covers white noise with margin (overflow is flagged, never silent)
    int cap = (int)((long long)L.w * L.h / 8);
    if (cap < 4096) cap = 4096;
    cap = (cap + 15) & ~15;   // score bytes of a level start 16-byte aligned
    L.cand_cap = cap;
    L.cand_off = cand;
    cand += cap;
    // the output-sensitive pass lists the maxima at or above the level's threshold -- about a thousand at 1280 x 720
    // against ~20 000 maxima in all --: the primary list is an eighth of the full one (a list that overflows all the same
    // sends its level to the dense pass, k_fast_check)
    int pcap = (int)((long long)L.w * L.h / 64);
    if (pcap < 4096) pcap = 4096;
    pcap = (pcap + 15) & ~15;
    // small levels keep their full capacity (little memory, and small levels are the ones that take the dense pass:
    // their thresholds rest on few corners): only lists above 16 K entries are cut
    if (pcap > cap || cap <= 16384) pcap = cap;
    prim_.cap[l] = pcap;
    prim_.off[l] = prim;
    prim += pcap;
    L.s1_off = s1;
    s1 += kS1Cap;
    // FAST tiles cover only the pixels runByImageBorder(31) can keep, [31, w-31) x [31, h-31) (17 % fewer tiles on
    // the 720p pyramid than tiling the whole level); levels without such pixels get no tiles
    L.tiles_x = L.w > 2 * kEdge ? (L.w - kEdge - kTileX0 + TW - 1) / TW : 0;
    L.tiles_y = L.h > 2 * kEdge ? (L.h - 2 * kEdge + TH - 1) / TH : 0;
    if (L.tiles_x == 0 || L.tiles_y == 0) L.tiles_x = L.tiles_y = 0;
    L.tile_base = tiles;
    tiles += L.tiles_x * L.tiles_y;
    // walker strips over the whole level: 256-px windows wk_px apart ((nx - 1) * wk_px + 256 >= w), wk_rows owned rows
    L.wk_nx = L.w > 256 ? (L.w - 256 + kWkMaxPx - 1) / kWkMaxPx + 1 : 1;
    L.wk_px = L.wk_nx > 1 ? (((L.w - 256 + L.wk_nx - 2) / (L.wk_nx - 1)) + 3) & ~3 : kWkMaxPx;
    const int rows_target = max_slots >= kWkTallMinSlots ? kWkRowsTall : kWkRowsShort;
    L.wk_ny = (L.h + rows_target - 1) / rows_target;
    L.wk_rows = (((L.h + L.wk_ny - 1) / L.wk_ny) + 3) & ~3;
    L.wk_ny = (L.h + L.wk_rows - 1) / L.wk_rows;
    L.wk_base = strips;
    strips += L.wk_nx * L.wk_ny;
    L.wk_fused = 0;
    L.tab_yemit = L.tab_xstrip = 0;
    if (L.tiles_x * L.tiles_y > g.max_level_tiles) g.max_level_tiles = L.tiles_x * L.tiles_y;
    // sample lattice of tau_unit: about 4096 pixels of the kept region in runs of 4 (one dword), rows sparser than
    // columns (a sampled pixel touches 7 rows); samp_sx counts dwords and is odd, so that block textures with
    // power-of-two periods are not aliased
    L.samp_sx = L.samp_sy = 1;
    L.samp_rows = L.samp_cols = 0;
    if (L.tiles_x > 0) {
      const int rh = L.h - 2 * kEdge;
      const int n_dw = (L.w - kEdge - ((kEdge + 3) & ~3)) >> 2;          // aligned dwords fully inside [31, w - 31)
      const double s2 = (double)n_dw * rh / (double)kTauSites;          // (dword, row) sites per sampled run
      int sx = (int)(sqrt(s2 > 1.0 ? s2 : 1.0) / 4.0);
      sx = (sx < 1 ? 1 : sx) | 1;
      int sy = (int)(s2 / sx + 0.5);
      sy = sy < 1 ? 1 : sy;
      L.samp_sx = sx;
      L.samp_sy = sy;
      L.samp_rows = (rh + sy - 1) / sy;
      L.samp_cols = n_dw > 0 ? (n_dw + sx - 1) / sx : 0;
      if (L.samp_cols == 0) L.samp_rows = 0;
    }
    L.tab_off = tab;
    if (l > 0) {
      tab += 12 * ((L.w + 3) >> 2) + ((L.h + 3) & ~3);   // 3 x uint4 per group of 4 columns, one dword per row
      L.tab_yemit = tab;                                 // one dword per SOURCE row (level l - 1)
      tab += (g.lv[l - 1].h + 3) & ~3;
      L.tab_xstrip = tab;                                // first output group of each walker strip of level l - 1, + end
      tab += (g.lv[l - 1].wk_nx + 1 + 3) & ~3;
    }
  }
  g.pyr_bytes = (pix + 255) & ~255ll;
  g.cand_total = cand;
  g.prim_total = prim;
  g.s1_total = s1;
  g.total_tiles = tiles;
  g.total_strips = strips;

  std::vector<uint32_t> htab(tab > 0 ? tab : 1, 0u);
  for (int l = 1; l < g.nlevels; l++) {
    const OrbLevelInfo& L = g.lv[l];
    const int groups = (L.w + 3) >> 2;
    std::vector<uint32_t> xt(4 * groups);
    make_table(g.lv[l - 1].w, L.w, xt.data());
    for (int x = L.w; x < 4 * groups; x++) xt[x] = xt[L.w - 1];   // pad entries: the last pixel again
    // k_resize<true> reads pixels 4g .. 4g+2 from the dword pair of pixel 4g: needs tap offset (rel. to that pair) <= 6
    bool shared = !resize_generic_;
    for (int x = 0; x < 4 * groups; x += 4)
      for (int k = 1; k < 3; k++)
        if ((xt[x + k] & 0xFFFFu) - (xt[x] & 0xFFFCu) > 6u) shared = false;
    resize_shared_[l] = shared;
    uint32_t* xsel = htab.data() + L.tab_off;
    uint32_t* xwxp = xsel + 4 * groups;
    uint32_t* xoff = xwxp + 4 * groups;
    for (int x = 0; x < 4 * groups; x++) {
      const uint32_t cx = xt[x] & 0xFFFFu, wx1 = xt[x] >> 16;
      const uint32_t base = ((shared && (x & 3) < 3) ? xt[x & ~3] : cx) & 0xFFFCu;   // byte offset of the aligned dword pair
      xoff[x] = base;
      xsel[x] = 0x0c010c00u + (cx - base) * 0x00010001u;   // pair bytes (cx - base, cx - base + 1) -> u16 lanes, zeros between
      xwxp[x] = (256u - wx1) | (wx1 << 16);
    }
    uint32_t* ytab = xoff + 4 * groups;
    make_table(g.lv[l - 1].h, L.h, ytab);
    // ---- tables of the fused form (the walker of level l - 1 makes this level): valid if every source row is the
    // upper tap of at most one output row, the shared-pair column form holds, and every strip's groups fit its window
    const OrbLevelInfo& Ls = g.lv[l - 1];
    bool fused = shared && Ls.wk_rows <= kWkMaxRows && Ls.h < 65536;
    uint32_t* yemit = htab.data() + L.tab_yemit;
    uint32_t* xstrip = htab.data() + L.tab_xstrip;
    for (int y = 0; y < L.h; y++) {
      const uint32_t sy = ytab[y] & 0xFFFFu, w1 = ytab[y] >> 16;
      if (sy >= (uint32_t)Ls.h || yemit[sy] != 0u || w1 > 256u) { fused = false; break; }
      yemit[sy] = 0x80000000u | (w1 << 16) | (uint32_t)y;
    }
    // group gq belongs to the last strip whose window starts at or before its first pair
    for (int c = 0; c <= Ls.wk_nx; c++) xstrip[c] = (uint32_t)groups;
    for (int gq = groups - 1; gq >= 0; gq--) {
      const uint32_t b0 = xoff[4 * gq], b3 = xoff[4 * gq + 3];
      int c = (int)(b0 / (uint32_t)Ls.wk_px);
      if (c > Ls.wk_nx - 1) c = Ls.wk_nx - 1;
      const uint32_t xs = (uint32_t)(c * Ls.wk_px);
      // every tap that counts lies inside the 256-px window; the second dword of a pair may stick out of it (it then
      // holds only weight-0 taps: the replicated last pixel) -- the read stays inside the wave's LDS block
      if (b0 < xs || b3 < b0 || b3 + 4u > xs + 256u) fused = false;
      for (int k = 0; k < 4; k++) {
        const uint32_t cx = xt[4 * gq + k] & 0xFFFFu, w1 = xt[4 * gq + k] >> 16;
        if (cx < xs || cx >= xs + 256u || (cx + 1u >= xs + 256u && w1 != 0u)) fused = false;
      }
      for (int cc = 0; cc <= c; cc++) xstrip[cc] = (uint32_t)gq;       // groups are monotone in b0: the last write is the first group
    }
    for (int c = 0; c < Ls.wk_nx; c++)
      if (xstrip[c + 1] < xstrip[c] || xstrip[c + 1] - xstrip[c] > 64u) fused = false;
    if (xstrip[0] != 0u) fused = false;
    if (Ls.wk_nx > kWkMaxNx) fused = false;
    for (int c = 0; c <= kWkMaxNx; c++) g.lv[l].wk_xg[c] = (int)xstrip[c < Ls.wk_nx ? c : Ls.wk_nx];
    g.lv[l].wk_fused = fused ? 1 : 0;
  }
  const size_t S = (size_t)max_slots, Wk = (size_t)work_frames_;
  MSF_HIP_TRY(hipMalloc(&d_pyr_, Wk * g.pyr_bytes));
  MSF_HIP_TRY(hipMalloc(&d_tab_, htab.size() * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemcpy(d_tab_, htab.data(), htab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  MSF_HIP_TRY(hipMalloc(&d_cand_cnt_, Wk * kOrbLevels * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMalloc(&d_tau_, 2 * Wk * kOrbLevels * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_tau_, 0, 2 * Wk * kOrbLevels * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMalloc(&d_redo_, 2 * (1 + Wk * kOrbLevels) * sizeof(uint32_t)));   // two queues: dense pass, dense pass again from the pool
  MSF_HIP_TRY(hipMalloc(&d_qstat_, Wk * kOrbLevels * kQStat * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_qstat_, 0, Wk * kOrbLevels * kQStat * sizeof(uint32_t)));
  {
    // candidate arrays: a primary region per work row + the pool.  The pool serves (a) the (frame, level)s that take the
    // dense second pass -- 0 to 8 of 2 048 in the measured batches; sized for an eighth of all of them at full capacity,
    // at least 64 level-0 lists -- and (b) dense CALLS, whose frames lay their full-capacity regions over it: every row
    // with MSF_FLAG_FAST_DENSE, at most stream_min_frames - 1 frames otherwise.
    const long long dense_rows = force_tau_ == kFastT ? (long long)Wk : std::min<long long>((long long)Wk, std::max(stream_min_frames_ - 1, 0));
    long long redo_entries = std::max<long long>((long long)Wk * g.cand_total / 8, 64ll * g.lv[0].cand_cap);
    redo_entries = std::min<long long>(redo_entries, (long long)Wk * g.cand_total);
    // (MSF_ORB_POOL_ENTRIES, tests: a pool too small for what a batch needs must end in MSF_ERR_CAPACITY, never in a
    // short list; batches that force most levels through the dense pass ask for more)
    if (const char* e = getenv("MSF_ORB_POOL_ENTRIES")) {
      const long long v = atoll(e);
      if (v >= 0) redo_entries = std::min<long long>(v, (long long)Wk * g.cand_total);
    }
    const long long pool_entries_ = std::max(std::max<long long>(redo_entries, 16), dense_rows * g.cand_total);
    g.pool_base = (long long)Wk * g.prim_total;
    const long long total = g.pool_base + pool_entries_;
    if (total >= (1ll << 32)) return "ORB candidate arrays exceed 2^32 entries: reduce max_batch_pairs";
    g.pool_entries = (unsigned)pool_entries_;
    MSF_HIP_TRY(hipMalloc(&d_cand_, (size_t)total * sizeof(uint32_t)));
    MSF_HIP_TRY(hipMalloc(&d_cand_sc_, (size_t)total));
    MSF_HIP_TRY(hipMalloc(&d_cmap_, Wk * kOrbLevels * sizeof(uint2)));
    MSF_HIP_TRY(hipMemset(d_cmap_, 0, Wk * kOrbLevels * sizeof(uint2)));
    MSF_HIP_TRY(hipMalloc(&d_pool_cnt_, 16));
    MSF_HIP_TRY(hipMemset(d_pool_cnt_, 0, 16));
  }
  MSF_HIP_TRY(hipMalloc(&d_walk_abort_, 16));
  MSF_HIP_TRY(hipMemset(d_walk_abort_, 0, 16));
  MSF_HIP_TRY(hipHostMalloc(&h_walk_abort_, 16, hipHostMallocDefault));
  h_walk_abort_[0] = h_walk_abort_[1] = 0u;
  MSF_HIP_TRY(hipMalloc(&d_s1_cnt_, Wk * kOrbLevels * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMalloc(&d_s1_, Wk * g.s1_total * sizeof(uint4)));
  MSF_HIP_TRY(hipMalloc(&d_kp_, S * kKpCap * sizeof(msf_keypoint)));
  MSF_HIP_TRY(hipMalloc(&d_desc_, S * kKpCap * 32));
  MSF_HIP_TRY(hipMalloc(&d_kp_cnt_, S * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMalloc(&d_status_, S * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMalloc(&d_qres_, (size_t)kSplitMaxPairs * kKpCap * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMalloc(&d_done_, (size_t)kSplitMaxPairs * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_done_, 0, (size_t)kSplitMaxPairs * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_kp_cnt_, 0, S * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_status_, 0, S * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_cand_cnt_, 0, Wk * kOrbLevels * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemset(d_s1_cnt_, 0, Wk * kOrbLevels * sizeof(uint32_t)));
  MSF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), k_orb_bit_pattern_31, 1024));
  {
    std::vector<uint32_t> disc(2 * kDiscTasks, 0u);   // padding tasks: zero weights
    for (int v = -15; v <= 15; v++) {
      const int dmax = g.umax[v < 0 ? -v : v];
      for (int k = 0; k < 8; k++) {
        uint32_t wu = 0, w1 = 0;
        for (int b = 0; b < 4; b++) {
          const int u = -16 + 4 * k + b;
          if (u >= -dmax && u <= dmax) {
            wu |= (uint32_t)(uint8_t)(int8_t)u << (8 * b);
            w1 |= (uint32_t)(uint8_t)(int8_t)v << (8 * b);
          }
        }
        disc[2 * ((v + 15) * 8 + k)] = wu;
        disc[2 * ((v + 15) * 8 + k) + 1] = w1;
      }
    }
    MSF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_disc), disc.data(), sizeof(uint32_t) * 2 * kDiscTasks));
  }
  if (profile_) {
    for (auto& set : evr_)
      for (auto& e : set.ev) MSF_HIP_TRY(hipEventCreate(&e));
    ev_ok_ = true;
  }
  return "";
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* a, uint32_t* b, uint32_t v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a[i] = v; b[i] = v; }
}

hipError_t OrbPipeline::extract(const FrameSrc& src, int n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (src.slot0 < 0 || src.slot0 + n > max_slots_ || n > work_frames_) return hipErrorInvalidValue;
  last_src_ = src;
  last_n_ = n;
  // A unit of an earlier walker launch gave up a bounded wait (its frames were flagged): the one-launch form rests on
  // workgroups being started in index order, which is observed, not promised.  From now on this handle launches one level
  // at a time (every wait is then met at once); the word arrives with the asynchronous copy behind each launch, so this
  // may take effect a few calls after the one that failed.
  if (h_walk_abort_ && !walk_per_level_ && __atomic_load_n(&h_walk_abort_[1], __ATOMIC_RELAXED) != 0u) {
    walk_per_level_ = true;
    degraded_note_ = true;
  }
  if (test_stall_frame_ >= 0 && test_stall_calls_++ > 0) test_stall_frame_ = -1;   // the test hook stalls the first call only
  if (ev_ok_) ev_begin_call();
  const hipError_t e = extract_range(src, n, st, ev_ok_ ? ev_ : nullptr);
  if (ev_ok_) ev_extract_pending_ = true;
  return e;
}

// pyramid, FAST, selection, descriptors of frames src[0 .. n) on stream st; evs (or null): five events recorded at the
// stage boundaries (start, pyramid + FAST, check + redo, selection, descriptors done)
hipError_t OrbPipeline::extract_range(const FrameSrc& src, int n, hipStream_t st, hipEvent_t* evs) {
  const OrbGeometry& g = g_;
  hipError_t e;
  // Work rows: frame i of this call uses row i of the per-call arrays (pyramid, candidate and stage-1 lists, thresholds,
  // walker state), whatever feature slot its key points go to.  The kernels index everything by slot = slot0 + i, so the
  // per-call arrays are passed as bases moved back by slot0 rows (never dereferenced below row 0: slots start at slot0).
  const ptrdiff_t back = (ptrdiff_t)src.slot0;
  uint8_t* const d_pyr_ = this->d_pyr_ - back * g.pyr_bytes;
  uint32_t* const d_cand_cnt_ = this->d_cand_cnt_ - back * kOrbLevels;
  uint2* const d_cmap_ = this->d_cmap_ - back * kOrbLevels;     // (its entries are absolute offsets into d_cand_ / d_cand_sc_)
  uint32_t* const d_qstat_ = this->d_qstat_ - back * kOrbLevels * kQStat;
  uint32_t* const d_s1_cnt_ = this->d_s1_cnt_ - back * kOrbLevels;
  uint4* const d_s1_ = this->d_s1_ - back * g.s1_total;
  if ((e = hipMemsetAsync(this->d_cand_cnt_, 0, (size_t)n * kOrbLevels * 4, st))) return e;
  if ((e = hipMemsetAsync(d_status_ + src.slot0, 0, (size_t)n * 4, st))) return e;
  if ((e = hipMemsetAsync(d_redo_, 0, 4, st))) return e;
  if ((e = hipMemsetAsync(d_redo_ + 1 + (size_t)work_frames_ * kOrbLevels, 0, 4, st))) return e;
  if (evs) hipEventRecord(evs[0], st);
  uint32_t* tau = d_tau_ - back * kOrbLevels;
  uint32_t* tau_first = d_tau_ + (ptrdiff_t)work_frames_ * kOrbLevels - back * kOrbLevels;
  // A call of a few frames (the single-pair MatchFrames, a key frame upload) is latency-bound: a wave of the
  // streaming pass walks its strip in ~90 dependent steps, whereas the dense tile kernel is one short workgroup per
  // tile.  Such calls take k_resize + the dense kernel directly; the result is the same either way.
  const int force_tau = (force_tau_ == 0 && n < stream_min_frames_) ? kFastT : force_tau_;
  const bool dense = force_tau == kFastT;
  // where each (frame, level)'s candidate list lives in this call (k_cand_reset): its primary list, or -- a dense call --
  // the frame's full-capacity region in the pool
  if (dense && (long long)n * g.cand_total > (long long)g.pool_entries) return hipErrorInvalidValue;   // (cannot happen: see init)
  hipLaunchKernelGGL(k_cand_reset, dim3((n * kOrbLevels + 255) / 256), dim3(256), 0, st, g, prim_, this->d_cmap_, d_pool_cnt_, n, dense ? 1 : 0);
  const int dyn = (fast_two_part_ && force_tau == 0) ? 1 : 0;
  bool fused = fused_ && !dense && g.total_tiles > 0;
  for (int l = 1; l < g.nlevels; l++) fused = fused && g.lv[l].wk_fused != 0;
  auto launch_resize = [&](int l) {
    const OrbLevelInfo& L = g.lv[l];
    // band height: 8 output rows x 256 threads measured best (rth 4: 3.26 ms, 8: 2.71, 12: 2.77, 16: 2.74 per 2048
    // [r02, table-driven kernel: 12 or 16 rows on the small levels only: 2.65 vs 2.65-2.73, within run-to-run noise]
    // 720p frames; workgroup sizes chosen to fill whole passes of 4 px x 4 row tasks -- 320..512 threads -- were
    // slower, 2.99: more, smaller workgroups hide the stage-then-compute latency better than full lanes do)
    const int sw16 = (g.lv[l - 1].w + 16 + 15) & ~15;
    const int groups = (L.w + 3) >> 2;
    int rth = 8;
    const int threads = 256;
    while (rth > 4 && ((rth * 5 + 3) / 4 + 3) * sw16 > 60000) rth -= 4;
    const int lds_rows = (rth * 5 + 3) / 4 + 3;
    const uint32_t magic_n16 = (uint32_t)(0x100000000ull / (uint32_t)(sw16 >> 4)) + 1u;
    const uint32_t magic_groups = groups > 1 ? (uint32_t)(0x100000000ull / (uint32_t)groups) + 1u : 0u;
    hipLaunchKernelGGL(resize_shared_[l] ? k_resize<true> : k_resize<false>, dim3((L.h + rth - 1) / rth, n), dim3(threads),
                       (size_t)lds_rows * sw16 + 8 * kResizeMaxRows, st, g, src, d_pyr_, d_tab_, l, rth, lds_rows, magic_n16,
                       magic_groups);
  };
  // the units of levels [l_lo, l_hi]: per XCD ceil(n / 8) frames x (1 threshold unit + the level's strips), see k_walk
  auto launch_walk = [&](int l_lo, int l_hi, int chain, int resize_mask, int predict_pct) {
    long long per_frame = 0;
    for (int l = l_lo; l <= l_hi; l++) per_frame += 1 + g.lv[l].wk_nx * g.lv[l].wk_ny;
    const long long wgs = 8ll * ((n + 7) / 8) * per_frame;
    hipLaunchKernelGGL(k_walk, dim3((unsigned)wgs), dim3(64), 0, st, g, src, d_pyr_, d_tab_, tau, tau_first, d_qstat_,
                       d_cand_cnt_, d_cand_, d_cand_sc_, d_cmap_, d_redo_, d_redo_ + 1, d_status_, d_walk_abort_, l_lo, l_hi, n,
                       tau2_margin_pct_, dyn, force_tau, predict_pct, chain, resize_mask, test_stall_frame_);
  };
  last_fused_ = fused;
  if (!dense && g.total_tiles > 0) {
    // the walker's per-(frame, level) state starts from zero: histograms, counters, "threshold published" words
    if ((e = hipMemsetAsync(this->d_qstat_, 0, (size_t)n * kOrbLevels * kQStat * 4, st))) return e;
    if ((e = hipMemsetAsync(d_walk_abort_, 0, 4, st))) return e;      // word 0 only: word 1 is the sticky stall count
  }
  if (fused) {
    // ONE launch: the walker of level l - 1 makes level l and finds level l - 1's corners in one pass over its pixels;
    // the thresholds, the order of the levels and the refinement are dependencies between the launch's units (k_walk).
    // r03 ran this stage as 2 chains x 8 levels x (sampler launch + walker launch): a fifth of its wave-slot time was
    // the tails of those 32 launches.
    const int mask = (1 << (g.nlevels - 1)) - 1;          // every level but the last also makes the next one
    const int predict = dyn ? tau_predict_pct_ : 0;
    if (walk_per_level_) {
      for (int l = 0; l < g.nlevels; l++) launch_walk(l, l, 1, mask, predict);
    } else {
      launch_walk(0, g.nlevels - 1, 1, mask, predict);
    }
    // both words: this launch's abort word and the sticky stall count (read at the start of a later call; the count
    // survives however many calls are enqueued before the host looks)
    hipMemcpyAsync(h_walk_abort_, d_walk_abort_, 8, hipMemcpyDeviceToHost, st);
    if (evs) hipEventRecord(evs[1], st);
  } else {
    for (int l = 1; l < g.nlevels; l++) launch_resize(l);
    if (evs) hipEventRecord(evs[1], st);
    if (g.total_tiles > 0) {
      if (dense) {   // MSF_FLAG_FAST_DENSE / a call of a few frames: the plain detector over every tile, nothing to verify
        hipLaunchKernelGGL(k_fill_u32, dim3((n * kOrbLevels + 255) / 256), dim3(256), 0, st, tau + (size_t)src.slot0 * kOrbLevels,
                           tau_first + (size_t)src.slot0 * kOrbLevels, (uint32_t)kFastT, n * kOrbLevels);
        hipLaunchKernelGGL(k_fast, dim3((unsigned)g.total_tiles * (unsigned)n), dim3(kFastThreads), 0, st, g, src, d_pyr_,
                           d_cand_cnt_, d_cand_, d_cand_sc_, d_cmap_);
      } else {
        // every level exists: FAST-only strips, every level's first threshold from its own sample
        launch_walk(0, g.nlevels - 1, 0, 0, 0);
      }
    }
  }
  if (g.total_tiles > 0 && !dense) {
    hipLaunchKernelGGL(k_fast_check, dim3((unsigned)n * kOrbLevels), dim3(64), 0, st, g, src.slot0, n, tau, tau_first,
                       d_cand_cnt_, d_cand_sc_, d_cmap_, d_redo_, d_redo_ + 1, d_qstat_);
    // fixed grid (a multiple of 8: see the XCD-contiguous unit order), sized to what the batch could need
    long long units = (long long)n * kOrbLevels * g.max_level_tiles;
    unsigned grid = (unsigned)(units < 2048 ? units : 2048);
    grid = (grid + 7u) & ~7u;
    hipLaunchKernelGGL(k_fast_redo, dim3(grid), dim3(kFastThreads), 0, st, g, src, d_pyr_, d_redo_, d_redo_ + 1,
                       g.max_level_tiles, d_cand_cnt_, d_cand_, d_cand_sc_, d_cmap_);
    // a redone level that overflowed its primary list: once more, into a full-capacity region of the pool (both launches
    // find an empty queue in nearly every call)
    uint32_t* const redo2 = d_redo_ + 1 + (size_t)work_frames_ * kOrbLevels;
    hipLaunchKernelGGL(k_redo_overflow, dim3((n * kOrbLevels + 255) / 256), dim3(256), 0, st, g, src.slot0, d_cand_cnt_, d_cmap_,
                       d_pool_cnt_, d_status_, d_redo_, d_redo_ + 1, redo2, redo2 + 1);
    hipLaunchKernelGGL(k_fast_redo, dim3(grid), dim3(kFastThreads), 0, st, g, src, d_pyr_, redo2, redo2 + 1,
                       g.max_level_tiles, d_cand_cnt_, d_cand_, d_cand_sc_, d_cmap_);
  }
  if (evs) hipEventRecord(evs[2], st);
  hipLaunchKernelGGL(k_thr_harris, dim3(g.nlevels, n), dim3(dense ? 256 : 64), 0, st, g, src, d_pyr_,
                     d_cand_cnt_, d_cand_, d_cand_sc_, d_cmap_, d_s1_cnt_, d_s1_, d_status_, dense ? 1 : 0);
  if (!dense) {
    HarrisUnits hw;
    int nu = 0;
    for (int l = 0; l < kOrbLevels; l++) {
      hw.base[l] = nu;
      hw.n[l] = l < g.nlevels ? (2 * g.lv[l].quota * 5 / 4 + 63) / 64 : 0;
      if (l < g.nlevels && hw.n[l] < 1) hw.n[l] = 1;
      nu += hw.n[l];
    }
    hw.base[kOrbLevels] = nu;
    for (int l = g.nlevels; l < kOrbLevels; l++) hw.base[l] = nu;
    hw.base[g.nlevels] = nu;
    hipLaunchKernelGGL(k_harris_flat, dim3((nu + 3) / 4, n), dim3(256), 0, st, g, src, d_pyr_, hw, d_s1_cnt_, d_s1_);
  }
  hipLaunchKernelGGL(k_select, dim3(n), dim3(256), 0, st, g, src.slot0, d_s1_cnt_, d_s1_, d_kp_, d_kp_cnt_,
                     d_status_);
  if (evs) hipEventRecord(evs[3], st);
  {
    // 4 key points per workgroup pass: 8 workgroups per frame keep a big batch busy; a single pair (the drop-in call)
    // gets up to 128 so that its ~500 key points per frame are one pass instead of sixteen
    int bx = 2048 / n;
    bx = bx < 8 ? 8 : bx > 128 ? 128 : bx;
    auto kd = blur_sum256_ ? k_describe<true, true> : half_up_ ? k_describe<true, false> : k_describe<false, false>;
    hipLaunchKernelGGL(kd, dim3(bx, n), dim3(256), 0, st, g, src, d_pyr_, d_kp_, d_kp_cnt_, d_desc_);
  }
  if (evs) hipEventRecord(evs[4], st);
  return hipGetLastError();
}

hipError_t OrbPipeline::match(int n_pairs, const int32_t* d_slot_a, const int32_t* d_slot_b, float ratio,
                              msf_match* d_out, int cap, int32_t* d_n_out, hipStream_t st, int slot_base, int slot_limit) {
  if (n_pairs <= 0) return hipSuccess;
  const int limit = slot_limit > 0 && slot_limit < max_slots_ ? slot_limit : max_slots_;
  // train descriptors go through LDS in chunks of kTrainChunk; MSF_ORB_TRAIN_CHUNK shrinks the chunk so tests can
  // exercise the multi-chunk path with ordinary keypoint counts
  if (ev_ok_ && !ev_extract_pending_) {
    ev_begin_call();                      // a slot-pair match on its own is a call of its own
    hipEventRecord(ev_[4], st);
  }
  static const int chunk = [] {
    const char* e = getenv("MSF_ORB_TRAIN_CHUNK");
    const int v = e ? atoi(e) : kTrainChunk;
    return v >= 16 && v <= kTrainChunk ? v : kTrainChunk;
  }();
  if (n_pairs <= kSplitMaxPairs && d_qres_)
    hipLaunchKernelGGL(k_match_split, dim3(kSplitBlocks, n_pairs), dim3(256), 0, st, n_pairs, d_slot_a, d_slot_b, d_kp_,
                       d_kp_cnt_, d_desc_, d_status_, ratio, d_out, cap, d_n_out, chunk, d_qres_, d_done_, slot_base, limit);
  else
    hipLaunchKernelGGL(k_match, dim3(n_pairs), dim3(256), 0, st, n_pairs, d_slot_a, d_slot_b, d_kp_, d_kp_cnt_,
                       d_desc_, d_status_, ratio, d_out, cap, d_n_out, chunk, slot_base, limit);
  if (ev_ok_) {
    hipEventRecord(ev_[5], st);
    EvSet& set = evr_[ev_cur_];
    set.recorded = true;
    set.match_only = !ev_extract_pending_;   // a slot-pair match on its own: only the last interval is of this call
    ev_extract_pending_ = false;
  }
  return hipGetLastError();
}

void OrbPipeline::ev_begin_call() {
  if (evr_[ev_cur_].recorded) {
    ev_cur_ = (ev_cur_ + 1) % kEvRing;
    if (evr_[ev_cur_].recorded) ev_harvest(ev_cur_);   // the oldest set, kEvRing calls back
  }
  ev_ = evr_[ev_cur_].ev;
}

// adds the stage times of set i to the accumulators (waits for the set's last event)
void OrbPipeline::ev_harvest(int i) {
  EvSet& set = evr_[i];
  if (!set.recorded) return;
  set.recorded = false;
  if (hipEventSynchronize(set.ev[5]) != hipSuccess) return;
  for (int k = set.match_only ? 4 : 0; k < 5; k++) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, set.ev[k], set.ev[k + 1]) == hipSuccess) acc_ms_[k] += t;
  }
  if (set.match_only) acc_match_only_++;
  else acc_full_++;
}

int OrbPipeline::stage_times(const char** names, float* ms, int cap) {
  // fused extraction: the first stage is pyramid + FAST of all levels (the one walker launch, thresholds included), the
  // second the check and the dense redo.  Sums over the calls since the last query.
  const char* kNames[5] = {last_fused_ ? "pyramid_fast" : "pyramid", "fast_nms", "select_harris", "orient_describe", "match"};
  if (!ev_ok_) return 0;
  for (int k = 0; k < kEvRing; k++) ev_harvest((ev_cur_ + 1 + k) % kEvRing);      // oldest first
  if (acc_full_ + acc_match_only_ == 0) return 0;
  int n = 0;
  for (int i = acc_full_ ? 0 : 4; i < 5 && n < cap; i++, n++) {
    names[n] = kNames[i];
    ms[n] = acc_ms_[i];
  }
  for (auto& a : acc_ms_) a = 0.f;
  acc_full_ = acc_match_only_ = 0;
  return n;
}

int OrbPipeline::debug_get(int what, int slot, int level, void* host_out, size_t cap, size_t* n_bytes,
                           std::string* err) {
  const OrbGeometry& g = g_;
  auto fail = [&](const char* m) { *err = m; return (int)MSF_ERR_INVALID_ARG; };
  auto copy_out = [&](const void* dsrc, size_t bytes) -> int {
    *n_bytes = bytes;
    const size_t n = bytes < cap ? bytes : cap;
    if (n && hipMemcpy(host_out, dsrc, n, hipMemcpyDeviceToHost) != hipSuccess) {
      *err = "hipMemcpy failed in debug_get";
      return (int)MSF_ERR_HIP;
    }
    return 0;
  };
  if (hipDeviceSynchronize() != hipSuccess) { *err = "hipDeviceSynchronize failed"; return MSF_ERR_HIP; }
  if (what == MSF_DBG_LEVEL_SIZES) {
    int32_t v[kOrbLevels][4];
    for (int l = 0; l < kOrbLevels; l++) { v[l][0] = g.lv[l].w; v[l][1] = g.lv[l].h; v[l][2] = g.lv[l].pitch; v[l][3] = g.lv[l].quota; }
    *n_bytes = sizeof(v);
    memcpy(host_out, v, sizeof(v) < cap ? sizeof(v) : cap);
    return 0;
  }
  if (what == MSF_DBG_WALK_MODE) {
    uint32_t stalls = 0;
    if (d_walk_abort_ && hipMemcpy(&stalls, d_walk_abort_ + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) {
      *err = "hipMemcpy failed in debug_get";
      return (int)MSF_ERR_HIP;
    }
    const int32_t v[2] = {walk_per_level_ || stalls != 0u ? 1 : 0, (int32_t)stalls};
    *n_bytes = sizeof(v);
    memcpy(host_out, v, sizeof(v) < cap ? sizeof(v) : cap);
    return 0;
  }
  if (slot < 0 || slot >= max_slots_) return fail("slot out of range");
  // key points and descriptors live per slot; everything else per work row of the LAST extraction
  const int row = slot - last_src_.slot0;
  if (what != MSF_DBG_KEYPOINTS && what != MSF_DBG_DESCRIPTORS && (row < 0 || row >= last_n_))
    return fail("only the frames of the last extraction have a pyramid / candidate lists (per-call workspace)");
  switch (what) {
    case MSF_DBG_LEVEL_PIXELS: {
      if (level < 1 || level >= g.nlevels) return fail("level must be 1..7 (level 0 is the input frame)");
      return copy_out(d_pyr_ + (size_t)row * g.pyr_bytes + g.lv[level].pix_off, (size_t)g.lv[level].pitch * g.lv[level].h);
    }
    case MSF_DBG_FAST_CANDS: {
      if (level < 0 || level >= g.nlevels) return fail("bad level");
      uint32_t n = 0;
      uint2 cm = make_uint2(0u, 0u);        // where the list lives: its primary region or a pool block
      hipMemcpy(&n, d_cand_cnt_ + (size_t)row * kOrbLevels + level, 4, hipMemcpyDeviceToHost);
      hipMemcpy(&cm, d_cmap_ + (size_t)row * kOrbLevels + level, sizeof(cm), hipMemcpyDeviceToHost);
      if (n > cm.y) n = cm.y;
      std::vector<uint32_t> tk(n);
      std::vector<uint8_t> ts(n);
      if (n) {
        hipMemcpy(tk.data(), d_cand_ + (size_t)cm.x, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        hipMemcpy(ts.data(), d_cand_sc_ + (size_t)cm.x, n, hipMemcpyDeviceToHost);
      }
      // the sampled quarter of a two-part streaming pass also lists corners below the level's final threshold
      uint32_t tfin = 0;
      hipMemcpy(&tfin, d_tau_ + (size_t)row * kOrbLevels + level, 4, hipMemcpyDeviceToHost);
      std::vector<int32_t> o;
      o.reserve(n * 3);
      for (uint32_t i = 0; i < n; i++)
        if (tfin <= (uint32_t)kFastT || ts[i] >= tfin) {
          o.push_back(tk[i] & 0xFFFF); o.push_back(tk[i] >> 16); o.push_back(ts[i]);
        }
      *n_bytes = o.size() * 4;
      memcpy(host_out, o.data(), *n_bytes < cap ? *n_bytes : cap);
      return 0;
    }
    case MSF_DBG_FAST_TAU: {
      int32_t v[kOrbLevels][2];
      uint32_t t[2][kOrbLevels];
      hipMemcpy(t[0], d_tau_ + (size_t)row * kOrbLevels, sizeof(t[0]), hipMemcpyDeviceToHost);
      hipMemcpy(t[1], d_tau_ + ((size_t)work_frames_ + row) * kOrbLevels, sizeof(t[1]), hipMemcpyDeviceToHost);
      for (int l = 0; l < kOrbLevels; l++) { v[l][0] = (int32_t)t[0][l]; v[l][1] = (int32_t)t[1][l]; }
      *n_bytes = sizeof(v);
      memcpy(host_out, v, sizeof(v) < cap ? sizeof(v) : cap);
      return 0;
    }
    case MSF_DBG_STAGE1: {
      if (level < 0 || level >= g.nlevels) return fail("bad level");
      uint32_t n = 0;
      hipMemcpy(&n, d_s1_cnt_ + (size_t)row * kOrbLevels + level, 4, hipMemcpyDeviceToHost);
      std::vector<uint4> tmp(n);
      if (n) hipMemcpy(tmp.data(), d_s1_ + (size_t)row * g.s1_total + g.lv[level].s1_off, n * sizeof(uint4), hipMemcpyDeviceToHost);
      std::vector<msf_keypoint> o(n);
      for (uint32_t i = 0; i < n; i++) {
        o[i].lx = tmp[i].x & 0xFFFF; o[i].ly = tmp[i].x >> 16; o[i].x = (float)o[i].lx; o[i].y = (float)o[i].ly;
        memcpy(&o[i].response, &tmp[i].y, 4); o[i].angle = -1.f; o[i].octave = level; o[i].fast_score = tmp[i].z;
      }
      *n_bytes = o.size() * sizeof(msf_keypoint);
      memcpy(host_out, o.data(), *n_bytes < cap ? *n_bytes : cap);
      return 0;
    }
    case MSF_DBG_KEYPOINTS:
    case MSF_DBG_DESCRIPTORS: {
      uint32_t n = 0;
      hipMemcpy(&n, d_kp_cnt_ + slot, 4, hipMemcpyDeviceToHost);
      if (n > (uint32_t)kKpCap) n = kKpCap;
      if (what == MSF_DBG_KEYPOINTS) return copy_out(d_kp_ + (size_t)slot * kKpCap, n * sizeof(msf_keypoint));
      return copy_out(d_desc_ + (size_t)slot * kKpCap * 32, (size_t)n * 32);
    }
    default:
      return fail("unknown debug item for ORB");
  }
}

}  // namespace msf
