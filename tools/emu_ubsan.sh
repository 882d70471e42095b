#!/bin/bash
# CPU-only: rebuilds the lane emulator of the kernel core (tests/emu) with UBSan + bounds checking and
# runs the emulator test files against it (GPU sanitizers are not available on the pool).
set -e
cd "$(dirname "$0")/.."
cp tests/emu/libcarma_emu.so /tmp/libcarma_emu.backup.so 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -ffp-contract=off -mfma -fsanitize=undefined,bounds \
    -fno-sanitize-recover=undefined -o tests/emu/libcarma_emu.so tests/emu/emu_core.cpp
touch tests/emu/libcarma_emu.so
python -m pytest tests/test_emu_core.py tests/test_emu_sampler.py -x -q
rm -f tests/emu/libcarma_emu.so      # the next test run rebuilds the normal library
