#!/bin/bash
# Builds variants of the library beside the default build and runs the probe: all builds in ONE process on the same (X, Y) pairs.  The
# variants of round 6 were compile-time switches in row_epilogue.h / spmm.hip (the Y stores' cache policy: default / nt / sc1 / sc0 sc1 / sc0 /
# sc1 nt; the edge stream loaded non-temporally) — the winner, non-temporal Y stores, is in the tree and the switches are gone; to repeat the
# experiment put one back as -DNAME and list it in VARIANTS.
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root/cleora_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
bash build.sh > /dev/null 2>&1
libs=""
for k in ${VARIANTS:-}; do
  hipcc $FLAGS -D$k -c spmm.hip -o /tmp/spmm_p$k.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC obj/dxd_host.o /tmp/spmm_p$k.o obj/rowops.o obj/whiten.o obj/project_f16.o obj/eigh.o obj/hot.o obj/attention.o obj/comm.o obj/peer.o obj/sharded.o obj/colsharded.o obj/multi.o obj/stager.o obj/similarity.o obj/abi.o -ldl -lpthread -lrt -o /tmp/libcleora_hip_p$k.so || exit 1
  libs="$libs /tmp/libcleora_hip_p$k.so"
done
for rep in 1 2; do python "$root/scripts/r06/store_policy_probe.py" $libs 2>&1 | grep '^{'; echo "--"; done
