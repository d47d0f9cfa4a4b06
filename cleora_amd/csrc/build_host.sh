#!/bin/bash
# Builds libcleora_host.so (pure host C++: entity hashing, graph builder, bincode pickle format).
#   build_host.sh            the product library
#   build_host.sh sanitize   libcleora_host_san.so with AddressSanitizer + UndefinedBehaviorSanitizer: loaded by
#                            tests/test_host_sanitizers.py through CLEORA_HOST_LIB with libasan preloaded; never shipped
set -euo pipefail
cd "$(dirname "$0")"
if [ "${1:-}" = sanitize ]; then
  OUT=../libcleora_host_san.so
  if [ ! -f $OUT ] || [ cleora_host.cpp -nt $OUT ] || [ ../../include/cleora_host.h -nt $OUT ]; then
    g++ -O1 -g -std=c++17 -fPIC -shared -Wall -Wextra -ffp-contract=off -pthread -fsanitize=address,undefined \
        -fno-sanitize-recover=undefined -fno-omit-frame-pointer cleora_host.cpp -o $OUT
  fi
  echo "built $(realpath $OUT)"
  exit 0
fi
OUT=../libcleora_host.so
if [ ! -f $OUT ] || [ cleora_host.cpp -nt $OUT ] || [ ../../include/cleora_host.h -nt $OUT ]; then
  g++ -O2 -std=c++17 -fPIC -shared -Wall -Wextra -ffp-contract=off -pthread cleora_host.cpp -o $OUT
fi
echo "built $(realpath $OUT)"
