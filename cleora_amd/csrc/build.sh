#!/bin/bash
# Builds libcleora_hip.so for gfx950 (cross-compiles without a GPU).
#   -ffp-contract=off : the SpMM must keep the reference's separate f32 multiply + add
#                       (src/embedding.rs:80-82); never let the compiler form FMAs.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcleora_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
mkdir -p obj
pids=()
for f in spmm rowops whiten project_f16 eigh hot attention comm peer sharded colsharded multi stager similarity abi; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ project_common.h -nt obj/$f.o ] || [ row_epilogue.h -nt obj/$f.o ] || [ comm_internal.h -nt obj/$f.o ] || [ shm_barrier.h -nt obj/$f.o ] || [ ../../include/cleora_hip.h -nt obj/$f.o ]; then
    # whiten.hip: MFMA accumulators in the VGPR form — hipcc otherwise parks loop-carried accumulators in VGPRs and copies
    # them to AGPRs and back around every chunk of MFMAs (256 v_accvgpr moves per 64 MFMAs in the Gram kernel)
    extra=""; [ $f = whiten ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    hipcc $FLAGS $extra -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
# host-only math (the d x d step of the intermediate whitened iterations for small d): plain C++, multi-versioned for AVX2 / AVX-512
if [ ! -f obj/dxd_host.o ] || [ dxd_host.cpp -nt obj/dxd_host.o ]; then g++ -O3 -std=c++17 -fPIC -c dxd_host.cpp -o obj/dxd_host.o; fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/dxd_host.o obj/spmm.o obj/rowops.o obj/whiten.o obj/project_f16.o obj/eigh.o obj/hot.o obj/attention.o obj/comm.o obj/peer.o obj/sharded.o obj/colsharded.o obj/multi.o obj/stager.o obj/similarity.o obj/abi.o -ldl -lpthread -lrt -o $OUT
echo "built $(realpath $OUT)"
