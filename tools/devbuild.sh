#!/bin/bash
# development build: fewer template instantiations (fast compile), every translation unit, same object cache and lock as
# daqp_amd.build().  Usage: tools/devbuild.sh [extra hipcc flags]     (daqp_amd.build() rebuilds the full library afterwards:
# objects are keyed on their flags and a "dev build" library always counts as stale)
cd "$(dirname "$0")/.." || exit 1
python - "$@" <<'PY'
import sys
import daqp_amd._lib as L
L.build(force=False, verbose=True, extra_flags=["-DDAQP_AMD_FEW_VARIANTS", *sys.argv[1:]])
PY
