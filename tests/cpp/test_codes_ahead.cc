// CodesAhead (guetzli_amd/host/codes_ahead.h): the helper thread that replays the next ten
// coefficient steps of phase B on private copies and computes the Huffman code lengths the size
// model will ask for -- checked against the same steps applied in sequence on this thread.
// Random blocks, random steps (several on one block, "precious" coefficients kept), thousands of
// jobs in bursts with sleeps in between (the helper's arm / rest cycle), jobs that are dropped.
// Build: g++ -O2 -std=c++17 -pthread test_codes_ahead.cc ../../guetzli_amd/host/jpeg_writer.cc -lz
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <random>
#include <thread>
#include <vector>

#include "../../guetzli_amd/host/codes_ahead.h"

using namespace guetzli_amd;

int main() {
  std::mt19937 rng(12345);
  const int kBlocks = 64;
  int q[3][64];
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 64; ++k) q[c][k] = 1 + (int)(rng() % 12);
  // an "image": dequantised coefficients, multiples of q (what the driver's mirror holds)
  std::vector<int16_t> img(3 * kBlocks * 64);
  auto randomize_block = [&](int c, int b) {
    int16_t* blk = &img[((size_t)c * kBlocks + b) * 64];
    for (int k = 0; k < 64; ++k) {
      const int r = (int)(rng() % 100);
      const int mag = r < 55 ? 0 : r < 85 ? 1 + (int)(rng() % 3) : 1 + (int)(rng() % 200);
      blk[k] = (int16_t)((rng() & 1 ? -mag : mag) * q[c][k]);
    }
  };
  for (int c = 0; c < 3; ++c)
    for (int b = 0; b < kBlocks; ++b) randomize_block(c, b);
  SymbolHistogram histo[3];
  for (int c = 0; c < 3; ++c)
    for (int b = 0; b < kBlocks; ++b)
      AddBlockACSymbols(&img[((size_t)c * kBlocks + b) * 64], q[c], 1, &histo[c]);

  CodesAhead ca;
  long jobs = 0, dropped = 0, mismatches = 0;
  for (int burst = 0; burst < 60; ++burst) {
    ca.Arm();
    const int ncomp = burst % 5 == 4 ? 1 : 3;
    for (int it = 0; it < 80; ++it) {
      CodesAhead::Job& job = ca.job();
      job.codes = &EntropyCodes;
      job.ncomp = ncomp;
      for (int c = 0; c < 3; ++c) {
        job.histo[c] = histo[c];
        job.q[c] = q[c];
      }
      job.nsteps = 1 + (int)(rng() % CodesAhead::kMaxSteps);
      job.nblocks = 0;
      int slot_c[CodesAhead::kMaxSteps], slot_b[CodesAhead::kMaxSteps];
      for (int s = 0; s < job.nsteps; ++s) {
        const int c = (int)(rng() % ncomp);
        const int b = (int)(rng() % (rng() % 3 == 0 ? 2 : kBlocks));   // repeats on a block are common
        int slot = -1;
        for (int t = 0; t < job.nblocks; ++t)
          if (slot_c[t] == c && slot_b[t] == b) slot = t;
        if (slot < 0) {
          slot = job.nblocks++;
          slot_c[slot] = c;
          slot_b[slot] = b;
          memcpy(job.blocks[slot], &img[((size_t)c * kBlocks + b) * 64], sizeof(job.blocks[slot]));
        }
        CodesAhead::Step& st = job.steps[s];
        st.slot = slot;
        st.c = c;
        st.k = 1 + (int)(rng() % 63);
        st.newval = (int16_t)(rng() % 2 ? 0 : (int)(rng() % 9 - 4) * q[c][st.k]);
        st.keep = st.newval == 0 && rng() % 7 == 0;
      }
      // what the driver would do in sequence (on the image itself)
      std::vector<CodesAhead::Step> steps(job.steps, job.steps + job.nsteps);
      ca.Post();
      for (int s = 0; s < (int)steps.size(); ++s) {
        const int c = steps[s].c, b = slot_b[steps[s].slot];
        int16_t* blk = &img[((size_t)c * kBlocks + b) * 64];
        AddBlockACSymbols(blk, q[c], -1, &histo[c]);
        if (!steps[s].keep) blk[steps[s].k] = steps[s].newval;
        AddBlockACSymbols(blk, q[c], 1, &histo[c]);
      }
      uint8_t depths[3 * kHistoSize];
      memset(depths, 0xee, sizeof(depths));
      const int header = (int)EntropyCodes(histo, ncomp, depths);
      if (rng() % 11 == 0) {   // the stopping rule fired inside the window: the result is never looked at
        ca.Drop();
        ++dropped;
      } else {
        ca.Wait();
        ++jobs;
        bool ok = job.header == header && memcmp(job.depths, depths, (size_t)ncomp * kHistoSize) == 0;
        for (int c = 0; c < ncomp; ++c)
          ok = ok && memcmp(job.histo[c].counts, histo[c].counts, sizeof(histo[c].counts)) == 0 &&
               job.raw_bits[c] == HistogramRawBits(histo[c], &depths[c * kHistoSize]);
        if (!ok) ++mismatches;
      }
      if (rng() % 50 == 0) {   // an edit outside the steps: the statistics are rebuilt
        randomize_block((int)(rng() % 3), (int)(rng() % kBlocks));
        for (int c = 0; c < 3; ++c) {
          histo[c].Clear();
          for (int b = 0; b < kBlocks; ++b)
            AddBlockACSymbols(&img[((size_t)c * kBlocks + b) * 64], q[c], 1, &histo[c]);
        }
      }
    }
    ca.Rest();
    if (burst % 7 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));
  }
  printf("codes ahead: %ld jobs checked, %ld dropped, %ld mismatches\n", jobs, dropped, mismatches);
  return mismatches == 0 && jobs > 3000 ? 0 : 1;
}
