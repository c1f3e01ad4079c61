#!/bin/bash
# tuning variants of ONE translation unit linked into side-by-side libraries (daqp_amd/lib/variants/libdaqp_amd_<tag>.so, selected with
# DAQP_AMD_LIBRARY): tools/variants.sh <unit.hip> <tag1>:"<flags>" <tag2>:"<flags>" ...     (the other objects: the current full build)
cd "$(dirname "$0")/.." || exit 1
UNIT=$1; shift
python -c "import daqp_amd; daqp_amd.build()" || exit 1
mkdir -p daqp_amd/lib/variants
OBJS=""
for o in daqp_amd/lib/obj/*.hip.o; do [ "$(basename $o)" != "$UNIT.o" ] && OBJS="$OBJS $o"; done
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $flags -c daqp_amd/csrc/$UNIT -o daqp_amd/lib/variants/$UNIT.$tag.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared $OBJS daqp_amd/lib/variants/$UNIT.$tag.o -o daqp_amd/lib/variants/libdaqp_amd_$tag.so && echo "built $tag" ) &
done
wait
