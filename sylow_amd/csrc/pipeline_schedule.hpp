// pipeline_schedule.hpp -- the chunk schedule of the host pipelines (pipeline.hip), plain C++ so that tests/cpp/schedule_test.cpp can
// compile it with g++ on a box without a GPU.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace pipeline {
// Chunk schedule.  A launch costs ~1 ms beyond its share of the work however large it is (its wavefronts start in step and drain
// unevenly; measured on k_pairing: 2^16 elements 8.6 ms, 2^18 30.7 ms, 2^20 119.5 ms), and the first chunk's upload and the last
// chunk's download are the only copies nothing hides.  So: a geometric ramp -- base, 4 base, 16 base, ... (chunk k + 1's upload, ~8 ns
// per element even from pageable memory, hides behind chunk k's ~115 ns per element as long as it is at most ~4 times as large), capped at
// 32 base, the rest in one piece -- and, when the results are large (Gt values: 384 bytes per element), one base chunk at the END so that the
// exposed download is short.  Returns the chunk boundaries (k + 1 offsets); the largest chunk is *cmax.
inline std::vector<size_t> schedule(size_t n, size_t base, bool large_results, size_t* cmax) {
  // sizes of the ramp-down at the end (large results only): ..., 4 base, base -- chunk k's download (~15 ns per element) hides behind chunk
  // k + 1's kernels as long as that chunk is not much smaller than a quarter of it, and only the last, small download is exposed
  std::vector<size_t> down;
  size_t reserved = 0;
  if (large_results)
    for (size_t c = base; c <= 4 * base && reserved + c + base <= n / 2; c *= 4) { down.push_back(c); reserved += c; }
  std::vector<size_t> cut(1, 0);
  size_t pos = 0, c = base;
  const size_t body = n - reserved;
  const size_t cap = 32 * base;                         // bounded device blocks: at most 2^21 elements (2.4 GB) per chunk at the default base
  while (pos < body) {
    size_t m = std::min(c, body - pos);
    const size_t left = body - pos - m;
    if (left < c) {                                     // what would be left is smaller than this chunk:
      if (m + left <= cap) m += left;                   // take it along where the block bound allows,
      else m = (m + left + 1) / 2;                      // otherwise split the remainder in two (each <= cap)
    }
    pos += m;
    cut.push_back(pos);
    c = std::min(4 * c, cap);
  }
  for (size_t i = down.size(); i-- > 0;) { pos += down[i]; cut.push_back(pos); }
  size_t mx = 0;
  for (size_t i = 1; i < cut.size(); ++i) mx = std::max(mx, cut[i] - cut[i - 1]);
  *cmax = mx;
  return cut;
}
}  // namespace pipeline
