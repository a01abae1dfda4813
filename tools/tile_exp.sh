#!/bin/bash
# timing experiments on the tile kernel: variants of the library under ab/ (tools/build_variant.sh), kernel stats of each
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/ab/libsynthray_$v.so
  [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/synthpy_amd/libsynthray.so
  SYNTHRAY_LIB=$lib SYNTHRAY_F64_TILE=1 tools/kstats.sh exp_$v 2>&1 | head -3 | sed "s/^/$v: /"
done
