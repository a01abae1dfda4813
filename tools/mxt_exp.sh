#!/bin/bash
# timing experiments on the mixed tile kernel: library variants under ab/ (tools/build_variant.sh), "name[:geometry]"
for vg in "$@"; do
  v=${vg%%:*}; g=${vg#*:}; [ "$g" = "$vg" ] && g=12,16,4,4,256
  lib=$GRAFT_REPO_ROOT/ab/libsynthray_$v.so
  [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/synthpy_amd/libsynthray.so
  SYNTHRAY_LIB=$lib SYNTHRAY_MX_TILE=1 SYNTHRAY_TILE=$g STEPS=2 tools/kstats.sh mexp_$v --precision mixed 2>&1 | head -2 | sed "s/^/$vg: /"
done
