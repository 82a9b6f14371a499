#!/bin/bash
# Builds experimental variants of libmoby_hip.so (extra -D flags) into build/variants/ and, with
# "run", times each on the GPU box with tools/world_profile.py's plain launch.
#   tools/variants.sh build "tag1:-DFOO=1 -DBAR" "tag2:..."      (here, cross-compiling)
#   tools/variants.sh run                                          (on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/variants
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    tag=${spec%%:*}; flags=${spec#*:}
    echo "== $tag: $flags"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $flags \
      -Rpass-analysis=kernel-resource-usage -o $ROOT/build/variants/lib_$tag.so $ROOT/moby_amd/csrc/mh_capi.hip 2>&1 | \
      grep -A10 "Function Name: _ZN2mh5small15mh_k" | grep "VGPRs:\|Spill\|ScratchSize\|LDS Size" | sed 's/.*remark: *//' | tr '\n' ' '
    echo
  done
else
  for so in $ROOT/build/variants/lib_*.so; do
    echo "== $so"
    MOBY_HIP_LIB=$so python $ROOT/tools/world_profile.py 4096 200 2>&1 | grep "plain launch"
  done
fi
