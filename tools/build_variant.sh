#!/bin/bash
# Build an A/B variant of libatx.so into anemoi-transform_amd/lib/variants/ (git-ignored, travels to the GPU box).
#   bash tools/build_variant.sh NAME [-DATX_KNOB=VALUE ...]          current sources with extra defines
#   bash tools/build_variant.sh NAME --rev <git-rev> [-D...]          the csrc/ + include/ of an earlier commit
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=$ROOT/anemoi-transform_amd/csrc
if [ "$1" == "--rev" ]; then
  REV=$2; shift 2
  TMP=$ROOT/gpurun_out/scratch/rev_$NAME
  rm -rf $TMP && mkdir -p $TMP/anemoi-transform_amd/csrc $TMP/include
  for f in $(git -C $ROOT ls-tree --name-only $REV anemoi-transform_amd/csrc/ include/); do git -C $ROOT show $REV:$f > $TMP/$f; done
  SRC=$TMP/anemoi-transform_amd/csrc
fi
mkdir -p $ROOT/anemoi-transform_amd/lib/variants
cd $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function "$@" \
  -o $ROOT/anemoi-transform_amd/lib/variants/libatx_$NAME.so *.hip
echo built $ROOT/anemoi-transform_amd/lib/variants/libatx_$NAME.so
