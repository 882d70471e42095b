#!/bin/bash
# Build the library, THEN send the tree to the GPU box: tools/gpu.sh [--timeout N] -- '<command>'
set -o pipefail
cd "$(dirname "$0")/.." && ./build.sh | tail -1 && exec /usr/local/graft/bin/gpurun "$@"
