#!/bin/bash
# Build libnnr_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libnnr_hip.so
FLAGS="${NNR_EXTRA_FLAGS} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed"
mkdir -p build
pids=()
compiled=""
kept=""
# incremental: a source is recompiled when it, common.h or the public header is newer than its object (NNR_BUILD_FORCE=1: everything);
# what was compiled and what was kept is printed, so a caller can see which it got
for f in gemm seq_plan lstm pool misc mhsa corpus dp gcn tape fuse sort; do
  [ -f $f.hip ] || continue
  if [ "${NNR_BUILD_FORCE}" = "1" ] || [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ ../../include/nnr_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
    compiled="$compiled $f"
  else
    kept="$kept $f"
  fi
done
for p in "${pids[@]}"; do wait $p; done
echo "compiled:${compiled:- (none)}; up to date:${kept:- (none)}"
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/*.o -ldl
echo "built $(realpath $OUT)"
