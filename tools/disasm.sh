#!/bin/bash
# Disassemble the gfx950 code object inside a built object file.  usage: tools/disasm.sh [pbhip32.o|pbhip64.o] [out.s]
# (llvm-objdump --offloading unbundles next to the input; the pieces are moved to /tmp/pbh_disasm)
set -euo pipefail
OBJ=${1:-pbhip32.o}
OUT=${2:-/tmp/pbh_disasm/${OBJ%.o}.s}
LLVM=/opt/rocm/lib/llvm/bin
SRC=$(cd "$(dirname "$0")/../pulsarbat_amd/csrc" && pwd)
mkdir -p /tmp/pbh_disasm "$(dirname "$OUT")"
( cd "$SRC" && $LLVM/llvm-objdump --offloading "$OBJ" > /dev/null && mv "$OBJ.0.hipv4-amdgcn-amd-amdhsa--gfx950" /tmp/pbh_disasm/${OBJ%.o}.co && rm -f "$OBJ".0.host-* )
$LLVM/llvm-objdump -d /tmp/pbh_disasm/${OBJ%.o}.co > "$OUT"
echo "$OUT"
