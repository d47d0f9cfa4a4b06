#!/bin/bash
# Builds libcleora_host.so (pure host C++: entity hashing, graph builder, bincode pickle format).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcleora_host.so
if [ ! -f $OUT ] || [ cleora_host.cpp -nt $OUT ] || [ ../../include/cleora_host.h -nt $OUT ]; then
  g++ -O2 -std=c++17 -fPIC -shared -Wall -Wextra -ffp-contract=off -pthread cleora_host.cpp -o $OUT
fi
echo "built $(realpath $OUT)"
