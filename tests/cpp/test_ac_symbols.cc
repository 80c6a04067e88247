// ReplaceCoeffACSymbols (guetzli_amd/host/jpeg_writer.cc: the symbols around ONE changed
// coefficient) against what it replaces -- AddBlockACSymbols(-1), the store, AddBlockACSymbols(+1):
// histograms and priced bits must be identical for every block shape (dense, sparse, empty, long
// zero runs with ZRL codes, a change at the first / last scan position, to and from zero).
#include <stdio.h>
#include <string.h>

#include <random>

#include "../../guetzli_amd/host/jpeg_writer.h"

using namespace guetzli_amd;

int main() {
  std::mt19937 rng(20260924);
  int q[64];
  uint8_t depth[kHistoSize];
  long cases = 0;
  for (int round = 0; round < 200000; ++round) {
    for (int i = 0; i < 64; ++i) q[i] = 1 + (int)(rng() % 40);
    for (int i = 0; i < kHistoSize; ++i) depth[i] = (uint8_t)(1 + rng() % 16);
    int16_t blk[64];
    const int density = (int)(rng() % 6);   // 0: empty ... 5: dense
    for (int i = 0; i < 64; ++i) {
      const bool nz = density == 5 ? true : density == 0 ? false : (int)(rng() % 16) < density * density;
      const int mag = 1 + (int)(rng() % (rng() % 4 == 0 ? 1000 : 6));
      blk[i] = nz ? (int16_t)((rng() & 1 ? -1 : 1) * mag * q[i]) : 0;
    }
    int k = 1 + (int)(rng() % 63);
    if (round % 7 == 0) k = kNaturalOrder[63];
    if (round % 11 == 0) k = kNaturalOrder[1];
    int newval = (rng() % 3 == 0 || blk[k] != 0) ? 0 : (int)((rng() & 1 ? -1 : 1) * (1 + rng() % 300) * q[k]);
    if (rng() % 9 == 0) newval = (int)((1 + rng() % 5) * q[k]);   // value to value
    const bool use_q = rng() % 5 != 0;
    SymbolHistogram a, b;
    for (int i = 0; i < 256; ++i) { const int w = (int)(rng() % 50) + 70; a.Add(i, w); b.Add(i, w); }
    uint32_t start_counts[kHistoSize];
    memcpy(start_counts, a.counts, sizeof start_counts);
    int64_t bits_a = 12345, bits_b = 12345;
    int16_t blk_a[64];
    memcpy(blk_a, blk, sizeof blk);
    AddBlockACSymbols(blk_a, use_q ? q : nullptr, -1, &a, depth, &bits_a);
    blk_a[k] = (int16_t)newval;
    AddBlockACSymbols(blk_a, use_q ? q : nullptr, 1, &a, depth, &bits_a);
    ReplaceCoeffACSymbols(blk, use_q ? q : nullptr, k, newval, &b, depth, &bits_b);
    ++cases;
    // ... and the same changes as a list (CoeffACSymbolChanges: what the search driver keeps of a step it
    // takes before the step's codes exist), applied and priced afterwards
    {
      SymbolHistogram c;
      for (int i = 0; i < kHistoSize; ++i) c.counts[i] = a.counts[i];   // = the result expected ...
      int16_t changes[kMaxCoeffACSymbolChanges];
      const int n = CoeffACSymbolChanges(blk, use_q ? q : nullptr, k, newval, changes);
      int64_t bits_c = bits_a;
      for (int j = 0; j < n; ++j) {   // ... undone: must give the statistics and the bits before the step
        const int symbol = (changes[j] > 0 ? changes[j] : -changes[j]) - 1, weight = changes[j] > 0 ? 1 : -1;
        c.Add(symbol, -weight);
        bits_c -= weight * (depth[symbol] + (symbol & 0xf));
      }
      if (n > kMaxCoeffACSymbolChanges || n < 0 || bits_c != 12345 ||
          memcmp(c.counts, start_counts, sizeof c.counts) != 0) {
        printf("MISMATCH (change list) round %d k %d old %d new %d n %d\n", round, k, blk[k], newval, n);
        return 1;
      }
    }
    if (memcmp(a.counts, b.counts, sizeof a.counts) != 0 || bits_a != bits_b) {
      printf("MISMATCH round %d k %d old %d new %d bits %lld vs %lld\n", round, k, blk[k], newval,
             (long long)bits_a, (long long)bits_b);
      return 1;
    }
  }
  printf("ac_symbols: ok (%ld cases)\n", cases);
  return 0;
}
