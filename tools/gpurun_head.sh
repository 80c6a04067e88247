#!/bin/bash
# gpurun with the commit stamped into the snapshot (which travels without .git): writes
# .gpurun_head = `git rev-parse HEAD` (+ "-dirty" when the tree has uncommitted changes), then runs
# gpurun with the arguments given.  tools/pmc_traffic_json.py and the session scripts copy the
# stamp into every profile they produce.   Usage: tools/gpurun_head.sh --timeout 900 -- '<command>'
cd "$(dirname "$0")/.."
h=$(git rev-parse HEAD)
git diff --quiet HEAD -- . ':!gpurun_out' || h="$h-dirty"
echo "$h" > .gpurun_head
exec /usr/local/graft/bin/gpurun "$@"
