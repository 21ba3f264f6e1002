#!/bin/bash
# usage: [ONLY="k_complex_f64 k_real_f32"] tools/build_variant.sh <name> <extra hipcc flags...>
#   -> kofft_amd/lib_<name>/libkofft_hip.so (A/B experiments; load with KOFFT_HIP_LIB or tools/ab_libs.py)
# ONLY: rebuild just these translation units with the extra flags and link them with the default build's other objects
# (run `make -C kofft_amd/csrc` first); without it every unit is rebuilt.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/kofft_variant_$NAME
rm -rf $W && mkdir -p $W/kofft_amd $W/include
cp -r $ROOT/kofft_amd/csrc $W/kofft_amd/csrc
cp $ROOT/include/*.h $W/include/
if [ -n "$ONLY" ]; then
    for u in $ONLY; do rm -f $W/kofft_amd/csrc/$u.o; done
    touch -c $W/kofft_amd/csrc/*.o   # the copied objects count as up to date
else
    rm -f $W/kofft_amd/csrc/*.o
fi
make -C $W/kofft_amd/csrc -j${JOBS:-8} EXTRA_HIPFLAGS="$*" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
mkdir -p $ROOT/kofft_amd/lib_$NAME
cp $W/kofft_amd/lib/libkofft_hip.so $ROOT/kofft_amd/lib_$NAME/
echo "built kofft_amd/lib_$NAME/libkofft_hip.so"
