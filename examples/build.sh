#!/bin/bash
# Builds examples/embed_file and examples/sharded_embed (plain C99, no Python, no torch) against the two in-tree libraries.
set -euo pipefail
cd "$(dirname "$0")"
LIB=$(realpath ../cleora_amd)
for prog in embed_file sharded_embed; do
  gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -I ../include $prog.c -o $prog \
      -L "$LIB" -lcleora_hip -lcleora_host -Wl,-rpath,"$LIB" -Wl,-rpath,/opt/rocm/lib
done
echo "built $(realpath embed_file) $(realpath sharded_embed)"
