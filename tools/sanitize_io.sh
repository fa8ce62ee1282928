#!/bin/bash
# Sanitizer runs of the host-only I/O code (writers, CSV reader, FASTA reader / packer) on the CPU:
# round trips on raw bit patterns and thousands of hostile files, the FASTA reader in forced multi-piece mode.
#   bash tools/sanitize_io.sh            AddressSanitizer + UBSan
#   SAN=thread bash tools/sanitize_io.sh ThreadSanitizer (thread pools of the writers, parallel FASTA pieces)
# GPU sanitizers are not available on the pool; this covers the code that parses untrusted text.
set -e
cd "$(dirname "$0")/.."
SAN=${SAN:-address,undefined}
mkdir -p /tmp/seekr_san
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fsanitize="$SAN" -fno-omit-frame-pointer \
    -Iinclude seekr_amd/csrc/io.hip seekr_amd/csrc/csv_read.hip seekr_amd/csrc/ctx.hip seekr_amd/csrc/comm.hip seekr_amd/csrc/pack.hip \
    tools/sanitize_io.cpp -o /tmp/seekr_san/drv -ldl -lpthread 2>/dev/null
ASAN_OPTIONS=detect_leaks=0 /tmp/seekr_san/drv
