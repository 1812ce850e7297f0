#!/bin/bash
# Builds a variant of libbrx.so with extra compiler flags for a same-box A/B:
#   tools/ab_build.sh NAME "-DBRX_XCD_ITEMS=0"   ->  br_amd/lib/ab/libbrx_NAME.so   (git-ignored, travels with gpurun)
# and is picked up with BRX_LIB_PATH=br_amd/lib/ab/libbrx_NAME.so python bench.py ...
# Different GPU boxes of the pool differ by up to ~10 % on the same code, so variants are compared inside one call.
set -e
cd "$(dirname "$0")/../br_amd/csrc"
name=$1; shift
bdir=build_ab_$name
mkdir -p $bdir ../lib/ab
srcs="brx_api brx_set brx_index brx_partbuild brx_scan brx_correct brx_onelane brx_pipeline brx_synth brx_exchange brx_devpool"
pids=()
for f in $srcs; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $f.hip -o $bdir/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""; for f in $srcs; do objs="$objs $bdir/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libbrx_$name.so $objs -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
rm -rf $bdir
echo "br_amd/lib/ab/libbrx_$name.so"
