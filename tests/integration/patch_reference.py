#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: INTEGRATION.md section 2 as compiled code.  Makes patched COPIES of two
files of the reference tree under tests/integration/_build/ref_patched/guetzli/ (git-ignored,
like oracle/_ref; nothing of the reference is committed here):

  comparator.h   + one virtual with a default, Comparator::ComputeAllBlockZeroingOrders(...):
                   "phase A for every block at once; false = not offered"
  processor.cc   SelectFrequencyMasking asks the comparator for the batched form first and
                   keeps its own per-block loop (processor.cc:554-590) as the fallback

The edits are anchored on single lines of the reference (asserted to be there exactly once), so
this file holds only what a maintainer would ADD.  `diff -u` of the result against the
reference is written next to the copies as batched_block_search.patch.

Usage: patch_reference.py <reference root> <output dir>"""
import os
import subprocess
import sys

HOOK_DECL = '''
  // Batched form of the per-block loop of Processor::SelectFrequencyMasking (phase A): the
  // zeroing order of EVERY block of the grid of comp_mask's last component, as that loop leaves
  // them in candidate_coeff_offsets / candidate_coeffs / candidate_coeff_errors.  Returns false
  // if the comparator does not offer it (the caller then runs the loop itself).
  virtual bool ComputeAllBlockZeroingOrders(
      const JPEGData& jpg, const OutputImage& img, uint8_t comp_mask, int lookahead,
      bool new_zeroing_model, std::vector<int>* candidate_coeff_offsets,
      std::vector<uint8_t>* candidate_coeffs, std::vector<float>* candidate_coeff_errors) {
    return false;
  }
'''

HOOK_CALL = '''  if (!comparator_->ComputeAllBlockZeroingOrders(
          jpg, *img, comp_mask, params_.zeroing_greedy_lookahead, params_.new_zeroing_model,
          &candidate_coeff_offsets, &candidate_coeffs, &candidate_coeff_errors)) {
'''


def insert_before(text, anchor, what, where):
    assert text.count(anchor) == 1, f"{where}: anchor {anchor!r} occurs {text.count(anchor)} times"
    return text.replace(anchor, what + anchor, 1)


def insert_after(text, anchor, what, where):
    assert text.count(anchor) == 1, f"{where}: anchor {anchor!r} occurs {text.count(anchor)} times"
    return text.replace(anchor, anchor + what, 1)


def main(ref, out):
    dst = os.path.join(out, "guetzli")
    os.makedirs(dst, exist_ok=True)
    # comparator.h: the declaration, in front of the class's last pure virtual
    src = open(os.path.join(ref, "guetzli", "comparator.h")).read()
    src = insert_before(src, "  // Returns a heuristic cutoff on block errors in the sense that we won't\n", HOOK_DECL.lstrip("\n") + "\n", "comparator.h")
    src = insert_after(src, '#include "guetzli/output_image.h"\n', '#include "guetzli/jpeg_data.h"\n', "comparator.h")
    open(os.path.join(dst, "comparator.h"), "w").write(src)
    # processor.cc: the per-block loop becomes the fallback of the batched call
    src = open(os.path.join(ref, "guetzli", "processor.cc")).read()
    src = insert_before(src, "  comparator_->StartBlockComparisons();\n", HOOK_CALL, "processor.cc")
    src = insert_after(src, "  candidate_coeff_offsets[num_blocks] = candidate_coeffs.size();\n", "  }\n", "processor.cc")
    open(os.path.join(dst, "processor.cc"), "w").write(src)
    with open(os.path.join(out, "batched_block_search.patch"), "w") as f:
        for name in ("comparator.h", "processor.cc"):
            f.write(subprocess.run(["diff", "-u", os.path.join(ref, "guetzli", name), os.path.join(dst, name)],
                                   capture_output=True, text=True).stdout)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
