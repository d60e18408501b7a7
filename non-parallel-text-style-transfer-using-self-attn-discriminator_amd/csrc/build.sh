#!/bin/bash
# Build libsad.so for gfx950 (MI355X), in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../_native"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function \
    ${SAD_EXTRA_FLAGS:-} \
    -o "$OUT/libsad.so" "$HERE/sad_api.hip"
echo "built $OUT/libsad.so"
