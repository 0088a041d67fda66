#!/bin/bash
# development: build a variant of libhaslr_hip.so into haslr_amd/lib_<name>/ (git-ignored; travels to the GPU box) from a copy of the sources with
# sed expressions applied to kernels/poa.hip (and extra hipcc flags): tools/dev_variant.sh name ['sed expr' ...] [-- -DFLAG ...]
set -e
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd); T=/tmp/v_$name
rm -rf $T; mkdir -p $T/haslr_amd; cp -r $R/include $T/; cp -r $R/haslr_amd/csrc $T/haslr_amd/; rm -rf $T/haslr_amd/csrc/.obj
extra=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; extra="$*"; break; fi
  sed -i "$1" $T/haslr_amd/csrc/kernels/poa.hip; shift
done
mkdir -p $R/haslr_amd/lib_$name
make -j8 -C $T/haslr_amd/csrc OUTLIB=$R/haslr_amd/lib_$name HX_EXTRA="$extra" $R/haslr_amd/lib_$name/libhaslr_hip.so 2>&1 | grep -E "error|Error" || true
ls -la $R/haslr_amd/lib_$name/libhaslr_hip.so
