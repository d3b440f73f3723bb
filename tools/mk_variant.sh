#!/bin/bash
# Build a variant of libsdso_hip.so for A/B runs on the GPU box:  tools/mk_variant.sh <name> "<EXTRA compiler flags>"
# -> ab_libs/<name>.so (git-ignored, travels with gpurun); select it with SDSO_LIB_PATH=$PWD/ab_libs/<name>.so (tools/ab_env.sh).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
tmp=/tmp/sdso_variant_$name
rm -rf $tmp; mkdir -p $tmp/stereo-dso-g2o_amd $tmp/include
cp -r $root/stereo-dso-g2o_amd/csrc $tmp/stereo-dso-g2o_amd/csrc
cp $root/include/*.h $tmp/include/
rm -f $tmp/stereo-dso-g2o_amd/csrc/*.o $tmp/stereo-dso-g2o_amd/csrc/*.so
make -s -j4 -C $tmp/stereo-dso-g2o_amd/csrc EXTRA="$*" 2>&1 | grep -E " error|Error" || true
mkdir -p $root/ab_libs
cp $tmp/stereo-dso-g2o_amd/csrc/libsdso_hip.so $root/ab_libs/$name.so
echo "built ab_libs/$name.so with EXTRA=$*"
