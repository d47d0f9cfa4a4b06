#!/bin/bash
# Builds examples/embed_file (plain C99, no Python, no torch) against the two in-tree libraries.
set -euo pipefail
cd "$(dirname "$0")"
LIB=$(realpath ../cleora_amd)
gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -I ../include embed_file.c -o embed_file \
    -L "$LIB" -lcleora_hip -lcleora_host -Wl,-rpath,"$LIB" -Wl,-rpath,/opt/rocm/lib
echo "built $(realpath embed_file)"
