#!/bin/bash
# Build two libraries HERE (no GPU needed) for a same-box A/B on the GPU box (tools/r06_gpu.sh insitu / gemmab):
#   tools/build/libmmpl_hip_prev.so = the plain product build (or, with --prev-from <git rev>, that revision's csrc)
#   tools/build/libmmpl_hip_new.so  = the tree built with the given flags ("" = the plain build of the working tree)
# usage: bash tools/prebuild_ab.sh [--prev-from <rev>] "<extra hipcc flags for the new library>"
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/build
prev_rev=""
if [ "$1" = "--prev-from" ]; then prev_rev=$2; shift; shift; fi
flags="$1"
if [ -n "$prev_rev" ]; then
  rm -rf /tmp/mmpl_prev && mkdir -p /tmp/mmpl_prev && git archive "$prev_rev" mmpl_amd include | tar -x -C /tmp/mmpl_prev
  (cd /tmp/mmpl_prev && python -m mmpl_amd.build > /tmp/mmpl_prev/build.log 2>&1) || { tail -5 /tmp/mmpl_prev/build.log; exit 1; }
  cp /tmp/mmpl_prev/mmpl_amd/lib/libmmpl_hip.so tools/build/libmmpl_hip_prev.so
else
  python -m mmpl_amd.build > /tmp/prebuild_prev.log 2>&1 || { tail -5 /tmp/prebuild_prev.log; exit 1; }
  cp mmpl_amd/lib/libmmpl_hip.so tools/build/libmmpl_hip_prev.so
fi
MMPL_EXTRA_HIPCC_FLAGS="$flags" python -m mmpl_amd.build > /tmp/prebuild_new.log 2>&1 || { tail -5 /tmp/prebuild_new.log; exit 1; }
cp mmpl_amd/lib/libmmpl_hip.so tools/build/libmmpl_hip_new.so
python -m mmpl_amd.build > /tmp/prebuild_restore.log 2>&1          # the tree is left with the plain product build
ls -la tools/build/*.so
