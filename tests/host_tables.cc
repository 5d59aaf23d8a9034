// Host check of the packed per-strategy tables of the product (jpegxl-rs_amd/csrc/jxl_dev.h) against the oracle's plain
// arrays (oracle/vardct.h).  Built and run by tests/test_host_tables.py (test infrastructure; not part of the product).
#include <cstdio>
#include "jxl_dev.h"
#include "vardct.h"
int main() {
  int bad = 0;
  for (uint32_t s = 0; s < 27; s++) {
    if (jxlhip::CoveredX(s) != jxlo::kCoveredX[s]) { printf("CoveredX(%u) = %u, oracle %u\n", s, jxlhip::CoveredX(s), jxlo::kCoveredX[s]); bad++; }
    if (jxlhip::CoveredY(s) != jxlo::kCoveredY[s]) { printf("CoveredY(%u) = %u, oracle %u\n", s, jxlhip::CoveredY(s), jxlo::kCoveredY[s]); bad++; }
    if (jxlhip::OrderBucket(s) != jxlo::kOrderBucket[s]) { printf("OrderBucket(%u) = %u, oracle %u\n", s, jxlhip::OrderBucket(s), jxlo::kOrderBucket[s]); bad++; }
    if (jxlhip::QuantKind(s) != jxlo::kQuantKind[s]) { printf("QuantKind(%u) = %u, oracle %u\n", s, jxlhip::QuantKind(s), jxlo::kQuantKind[s]); bad++; }
  }
  printf("%d mismatches\n", bad);
  return bad != 0;
}
