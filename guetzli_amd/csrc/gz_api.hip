// C-ABI implementation (include/guetzli_amd.h): per-image context, device plane arena,
// blur plans, and the kernel sequences for the block path and for
// ButteraugliComparator::Compare.  Host code only orchestrates; every per-pixel /
// per-block operation is in the gz_kernels_*.h kernels.
//
// Built by hipcc for gfx950 with -ffp-contract=off (guetzli_amd/build.py).  There is no
// CPU path: without a usable HIP device gz_create fails with GZ_E_NO_DEVICE.

#include "../../include/guetzli_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <sched.h>
#include <thread>
#include <utility>
#include <vector>

#include "gz_common.h"
#include "gz_kernels_block.h"
#include "gz_kernels_blur.h"
#include "gz_kernels_diff.h"
#include "gz_kernels_search.h"
#include "gz_kernels_entropy.h"
#include "gz_kernels_dctd.h"
#include "gz_kernels_downsample.h"
#include "gz_kernels_order.h"
#include "gz_kernels_rank.h"
#include "gz_host_weights.h"
#include "order_tables_generated.h"   // host-side csf/bias of order.inc

using namespace gz;

// The implementation by concern (one translation unit: the kernels are templates and static
// functions of the headers above, instantiated where they are launched):
#include "api/plans.h"          // blur plans, Malta normalisations, PsychoImage planes
#include "api/context.h"        // pools, gz_ctx, error macros, plane arena
#include "api/chain.h"          // the Compare chain's stages on three streams
#include "api/entry_context.h"  // gz_create / gz_destroy / ...
#include "api/entry_compare.h"  // block path + gz_compare*
#include "api/entry_phaseb.h"   // gz_order_* / gz_apply_*
#include "api/entry_entropy.h"  // gz_jpeg_*
#include "api/entry_frame.h"    // context-free transforms, 4:2:0 frame
#include "api/entry_search.h"   // gz_block_zeroing_orders*, gz_compare_blocks
#include "api/entry_probes.h"   // gz_probe_* (diagnostics)
