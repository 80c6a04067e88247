// CodeRefreshers (guetzli_amd/host/code_refresh.h): the hand-over of the size model's code refreshes
// to helper threads.  Streams of slowly drifting AC statistics go through 1..4 helpers with up to
// threads + 1 windows in flight, in bursts separated by Deactivate / Activate (the helpers sleep
// between phase B's iterations); every result must equal EntropyCodes / HistogramRawBits computed
// on the spot.  Built with -fsanitize=thread by the test: the protocol has no data race.
#include <stdio.h>
#include <string.h>

#include <random>
#include <vector>

#include "../../guetzli_amd/host/code_refresh.h"

using namespace guetzli_amd;

int main() {
  std::mt19937 rng(20260924);
  long checked = 0;
  for (int threads = 1; threads <= 4; ++threads) {
    CodeRefreshers cr(threads);
    SymbolHistogram h[3];
    for (int c = 0; c < 3; ++c)
      for (int i = 0; i < 256; ++i)
        if (rng() % 3) h[c].Add(i, (int)(rng() % (i < 32 ? 100000 : 500)) + 1);
    std::vector<CodeRefresh> expect(CodeRefreshers::kSlots);
    for (int burst = 0; burst < 40; ++burst) {
      if (burst % 3 != 2) cr.Activate();            // (every third burst relies on Submit's own wake-up)
      const int ncomp = burst % 5 == 4 ? 1 : 3;
      const long w0 = cr.NextWindow();
      const long windows = 1 + (long)(rng() % 60);
      long submitted = 0, waited = 0;
      while (waited < windows) {
        if (submitted < windows && submitted - waited <= threads) {
          for (int s = 0; s < 10; ++s) {             // ten "steps"
            const int c = (int)(rng() % 3), sym = (int)(rng() % 256);
            if (h[c].counts[sym] >= 4 && rng() % 2) h[c].Add(sym, -1); else h[c].Add(sym, 1);
          }
          CodeRefresh* in = cr.Input(w0 + submitted);
          memcpy(in->histo, h, sizeof(in->histo));
          in->ncomp = ncomp;
          CodeRefresh& e = expect[(w0 + submitted) % CodeRefreshers::kSlots];
          memcpy(e.histo, h, sizeof(e.histo));
          e.ncomp = ncomp;
          memset(e.depths, 0, sizeof e.depths);
          e.ac_header = (int)EntropyCodes(e.histo, ncomp, e.depths);
          for (int c = 0; c < 3; ++c)
            e.raw_bits[c] = c < ncomp ? HistogramRawBits(e.histo[c], &e.depths[c * kHistoSize]) : 0;
          cr.Submit(w0 + submitted);
          ++submitted;
          continue;
        }
        const CodeRefresh* r = cr.Wait(w0 + waited);
        const CodeRefresh& e = expect[(w0 + waited) % CodeRefreshers::kSlots];
        if (r->ac_header != e.ac_header || memcmp(r->raw_bits, e.raw_bits, sizeof e.raw_bits) != 0 ||
            memcmp(r->depths, e.depths, (size_t)ncomp * kHistoSize) != 0) {
          printf("MISMATCH threads %d burst %d window %ld\n", threads, burst, waited);
          return 1;
        }
        ++waited;
        ++checked;
      }
      cr.Deactivate();
    }
  }
  printf("code_refresh: ok (%ld refreshes)\n", checked);
  return 0;
}
