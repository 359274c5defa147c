// CPU-only check of the host pipelines' chunk schedule (sylow_amd/csrc/pipeline_schedule.hpp): the cuts cover [0, n) in order, no chunk
// is empty, and no chunk exceeds 32 * base elements (the bound the device blocks are sized by) -- for every n in a sweep that includes the
// sizes where the "take the remainder along" rule used to overshoot (n just below 64 * base past the ramp).
#include "../../sylow_amd/csrc/pipeline_schedule.hpp"

#include <cstdio>

int main() {
  const size_t bases[] = {1, 7, 128, size_t(1) << 16};
  size_t checked = 0;
  for (size_t base : bases)
    for (int large = 0; large < 2; ++large) {
      std::vector<size_t> ns;
      for (size_t n = 1; n <= 300; ++n) ns.push_back(n);
      for (size_t k = 1; k <= 200; ++k) { ns.push_back(k * base); ns.push_back(k * base + 1); if (k * base > 1) ns.push_back(k * base - 1); }
      for (size_t sh = 16; sh <= 25; ++sh) { ns.push_back(size_t(1) << sh); ns.push_back((size_t(1) << sh) - 1); ns.push_back((size_t(1) << sh) + 12345); }
      for (size_t n : ns) {
        size_t cmax = 0;
        const std::vector<size_t> cut = pipeline::schedule(n, base, large != 0, &cmax);
        if (cut.size() < 2 || cut.front() != 0 || cut.back() != n) { printf("FAIL cover n=%zu base=%zu\n", n, base); return 1; }
        size_t mx = 0;
        for (size_t i = 1; i < cut.size(); ++i) {
          if (cut[i] <= cut[i - 1]) { printf("FAIL order n=%zu base=%zu\n", n, base); return 1; }
          mx = std::max(mx, cut[i] - cut[i - 1]);
        }
        if (mx != cmax || cmax > 32 * base) { printf("FAIL bound n=%zu base=%zu cmax=%zu\n", n, base, cmax); return 1; }
        ++checked;
      }
    }
  printf("OK %zu schedules\n", checked);
  return 0;
}
