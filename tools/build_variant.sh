#!/bin/bash
# tools/build_variant.sh TAG "<extra hipcc flags>": builds build/variants/libmoby_hip_TAG.so from the same sources with extra
# -D flags (kernel experiments; select it with MOBY_HIP_LIB=... as moby_amd/_lib.py allows)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/variants/obj_$1
for f in $ROOT/moby_amd/csrc/*.hip; do
  o=$ROOT/build/variants/obj_$1/$(basename $f .hip).o
  # only the LCP translation unit depends on the block-solver switches: reuse the main build's objects for the rest
  if [ "$(basename $f)" = "mh_lcp_blk.hip" ] || [ "$(basename $f)" = "mh_lcp_blkw.hip" ] || [ "$(basename $f)" = "mh_lcp_blk1.hip" ] || [ "$(basename $f)" = "mh_lcp_blk2.hip" ] || [ "$(basename $f)" = "mh_lcp_blkx.hip" ] || [ "$(basename $f)" = "mh_lcp_blky.hip" ] || [ ! -f $ROOT/build/obj/$(basename $f .hip).o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $2 -c -o $o $f &
  else cp $ROOT/build/obj/$(basename $f .hip).o $o; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/variants/libmoby_hip_$1.so $ROOT/build/variants/obj_$1/*.o
echo built $ROOT/build/variants/libmoby_hip_$1.so
