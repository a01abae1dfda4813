#!/bin/bash
# PMC of the tile kernel for library variants under ab/
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/ab/libsynthray_$v.so
  [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/synthpy_amd/libsynthray.so
  SYNTHRAY_LIB=$lib SYNTHRAY_F64_TILE=1 tools/pmc_tile.sh v_$v k_trace_tile 2>&1 | sed "s/^/$v: /"
done
