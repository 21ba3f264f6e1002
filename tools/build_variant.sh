#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> kofft_amd/lib_<name>/libkofft_hip.so (A/B experiments; use with KOFFT_HIP_LIB)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/kofft_variant_$NAME
rm -rf $W && mkdir -p $W/kofft_amd $W/include
cp -r $ROOT/kofft_amd/csrc $W/kofft_amd/csrc
cp $ROOT/include/*.h $W/include/
rm -f $W/kofft_amd/csrc/*.o
make -C $W/kofft_amd/csrc -j${JOBS:-8} EXTRA_HIPFLAGS="$*" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
mkdir -p $ROOT/kofft_amd/lib_$NAME
cp $W/kofft_amd/lib/libkofft_hip.so $ROOT/kofft_amd/lib_$NAME/
echo "built kofft_amd/lib_$NAME/libkofft_hip.so"
