/*
 * Deterministic, integer-only synthetic frame pairs:
 * PCG32 stream seeded per pair, a coarse random u8 grid upsampled x8 (blocky
 * corners FAST fires on) or bilinearly (soft blobs), frame B a shifted crop of
 * the same canvas, independent +-8 noise per frame.  Host-only helper used by
 * tests/ and bench.py to make identical bytes for the CPU and GPU paths; it is
 * not part of the matcher hot path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t state, inc; } pcg32_t;

static uint32_t pcg32_next(pcg32_t* r) {
  uint64_t old = r->state;
  r->state = old * 6364136223846793005ULL + r->inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
}
static void pcg32_seed(pcg32_t* r, uint64_t seed, uint64_t seq) {
  r->state = 0u;
  r->inc = (seq << 1u) | 1u;
  pcg32_next(r);
  r->state += seed;
  pcg32_next(r);
}

#define MARGIN 32

static void crop_noise(const uint8_t* canvas, int cw, int ox, int oy, int w, int h,
                       pcg32_t* rng, int noise, uint8_t* out, int64_t stride) {
  for (int y = 0; y < h; y++) {
    const uint8_t* s = canvas + (size_t)(oy + y) * cw + ox;
    uint8_t* d = out + (size_t)y * stride;
    for (int x = 0; x < w; x++) {
      int v = s[x];
      if (noise > 0) v += (int)((pcg32_next(rng) >> 8) % (uint32_t)(2 * noise + 1)) - noise;
      d[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

/* mode 0: blocky (nearest x8); mode 1: smooth (bilinear x8); mode 2: blocky x16
 * |dx|, |dy| <= 32.  Returns 0 on success. */
int msf_synth_pair(uint64_t seed, int w, int h, int dx, int dy, int mode, int noise,
                   uint8_t* a, int64_t stride_a, uint8_t* b, int64_t stride_b) {
  if (w <= 0 || h <= 0 || dx < -MARGIN || dx > MARGIN || dy < -MARGIN || dy > MARGIN) return 1;
  int cell = mode == 2 ? 16 : 8;
  int cw = w + 2 * MARGIN, ch = h + 2 * MARGIN;
  int gw = cw / cell + 2, gh = ch / cell + 2;
  pcg32_t g, na, nb;
  pcg32_seed(&g, seed, 1);
  pcg32_seed(&na, seed, 2);
  pcg32_seed(&nb, seed, 3);
  uint8_t* grid = (uint8_t*)malloc((size_t)gw * gh);
  uint8_t* canvas = (uint8_t*)malloc((size_t)cw * ch);
  if (!grid || !canvas) { free(grid); free(canvas); return 2; }
  for (int i = 0; i < gw * gh; i++) grid[i] = (uint8_t)(pcg32_next(&g) >> 24);
  for (int y = 0; y < ch; y++)
    for (int x = 0; x < cw; x++) {
      int gx = x / cell, gy = y / cell;
      if (mode == 1) {
        int fx = x % cell, fy = y % cell;
        int p00 = grid[gy * gw + gx], p01 = grid[gy * gw + gx + 1];
        int p10 = grid[(gy + 1) * gw + gx], p11 = grid[(gy + 1) * gw + gx + 1];
        int top = p00 * (cell - fx) + p01 * fx, bot = p10 * (cell - fx) + p11 * fx;
        canvas[(size_t)y * cw + x] = (uint8_t)((top * (cell - fy) + bot * fy + cell * cell / 2) / (cell * cell));
      } else {
        canvas[(size_t)y * cw + x] = grid[gy * gw + gx];
      }
    }
  crop_noise(canvas, cw, MARGIN, MARGIN, w, h, &na, noise, a, stride_a);
  crop_noise(canvas, cw, MARGIN + dx, MARGIN + dy, w, h, &nb, noise, b, stride_b);
  free(grid);
  free(canvas);
  return 0;
}

/* The LoFTR known-answer pattern of SURVEY.md section 8c:
 * P(x,y) = ((x>>3)*37 + (y>>3)*101 + (x>>3)*(y>>3)*17) & 255, shifted by (sx, sy). */
void msf_synth_kat_pattern(int w, int h, int sx, int sy, uint8_t* out, int64_t stride) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int X = (x + sx) >> 3, Y = (y + sy) >> 3;
      out[(size_t)y * stride + x] = (uint8_t)((X * 37 + Y * 101 + X * Y * 17) & 255);
    }
}
