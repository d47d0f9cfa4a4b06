#!/bin/bash
# builds the measurement-only side kernels (not part of the product library)
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -shared -fPIC side_load.hip -o libside_load.so
echo built scripts/probes/libside_load.so
