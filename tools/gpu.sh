#!/bin/bash
# Build the library, THEN send the tree to the GPU box: tools/gpu.sh [--timeout N] -- '<command>'
cd "$(dirname "$0")/.." && ./build.sh | tail -1 && exec /usr/local/graft/bin/gpurun "$@"
