// tz_pattern.h -- the candidate pattern of TzSearch::FullpelDiamondSearch
// (inter_tz_search.cc:173-210) as data: for every index of the concatenated
// diamonds of ranges 1,2,4,...,256 the offset from the search centre, the one
// or two window bounds CheckCost1/2 test (:304-336), the `last_range_` value a
// hit records and the round (range index) it belongs to.  Built once on the
// host (xvcgpu_create) by the same function the device can call, so there is
// a single definition of the issue order.
#ifndef XVCGPU_TZ_PATTERN_H_
#define XVCGPU_TZ_PATTERN_H_

#include <stdint.h>

#ifdef __HIPCC__
#define TZ_HD __host__ __device__
#else
#define TZ_HD
#endif

enum { TZP_LEFT = -1, TZP_RIGHT = 1, TZP_UP = -3, TZP_DOWN = 3 };

struct TzCand {
  int16_t dx, dy;   // offset from the search centre
  int8_t d1, d2;    // bound(s) tested: TZP_* codes, d2 == 0 for CheckCost1
  uint8_t round;    // 0 for range 1, 1 for range 2, ...
  uint8_t pad;
  int16_t rng;      // value stored into last_range_ on a hit
  int16_t pad2;
};

#define TZ_MAX_ROUNDS 9  // ranges 1..256
#define TZ_MAX_CANDS 108 // 4 + 3*8 + 5*16

TZ_HD inline int tz_pattern_count(int range) {
  return range == 1 ? 4 : (range <= 8 ? 8 : 16);
}

// k-th candidate of the diamond of `range` in issue order.
TZ_HD inline TzCand tz_pattern_cand(int range, int k) {
  TzCand c;
  c.pad = 0;
  c.pad2 = 0;
  c.round = 0;
  int d1 = 0, d2 = 0, dx = 0, dy = 0, rng = range;
  const int dirs[4] = {TZP_UP, TZP_LEFT, TZP_RIGHT, TZP_DOWN};
  if (range == 1 || (range > 8 && k < 4)) {
    d1 = dirs[k];
    dx = d1 == TZP_LEFT ? -range : (d1 == TZP_RIGHT ? range : 0);
    dy = d1 == TZP_UP ? -range : (d1 == TZP_DOWN ? range : 0);
  } else if (range <= 8) {
    const int r2 = range >> 1;
    switch (k) {
      case 0: d1 = TZP_UP; dy = -range; break;
      case 1: d1 = TZP_UP; d2 = TZP_LEFT; dx = -r2; dy = -r2; rng = r2; break;
      case 2: d1 = TZP_UP; d2 = TZP_RIGHT; dx = r2; dy = -r2; rng = r2; break;
      case 3: d1 = TZP_LEFT; dx = -range; break;
      case 4: d1 = TZP_RIGHT; dx = range; break;
      case 5: d1 = TZP_DOWN; d2 = TZP_LEFT; dx = -r2; dy = r2; rng = r2; break;
      case 6: d1 = TZP_DOWN; d2 = TZP_RIGHT; dx = r2; dy = r2; rng = r2; break;
      default: d1 = TZP_DOWN; dy = range; break;
    }
  } else {
    const int i = 1 + ((k - 4) >> 2), q = (k - 4) & 3;
    const int r14 = i * (range >> 2), r34 = range - r14;
    d1 = (q < 2) ? TZP_UP : TZP_DOWN;
    d2 = (q & 1) ? TZP_RIGHT : TZP_LEFT;
    dx = (q & 1) ? r14 : -r14;
    dy = (q < 2) ? -r34 : r34;
  }
  c.dx = (int16_t)dx;
  c.dy = (int16_t)dy;
  c.d1 = (int8_t)d1;
  c.d2 = (int8_t)d2;
  c.rng = (int16_t)rng;
  return c;
}

// Fills table[0..TZ_MAX_CANDS) in issue order; returns the entry count.
inline int tz_pattern_build(TzCand *table) {
  int n = 0, round = 0;
  for (int range = 1; range <= 256; range *= 2, round++)
    for (int k = 0; k < tz_pattern_count(range); k++) {
      table[n] = tz_pattern_cand(range, k);
      table[n].round = (uint8_t)round;
      n++;
    }
  return n;
}

#endif  // XVCGPU_TZ_PATTERN_H_
