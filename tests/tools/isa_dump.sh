#!/bin/bash
# isa_dump.sh <outdir> [object glob ...] -- disassembles the gfx950 code objects inside the built objects of libavif_amd/csrc
# (default: the tiled-kernel families) into <outdir>/<object>.s, addresses and encodings stripped, and prints one checksum
# over all of them.  For refactors that must not change the device code: dump before, rebuild, dump after, then
#     for f in before/*.s; do cmp -s $f after/$(basename $f) || echo "differs: $f"; done
# and, where they differ, compare the opcode multisets (awk '{print $1}' | sort | uniq -c): a pure re-allocation of scalar
# registers shows the same multiset with s_* lines renamed (this is how round 2's tile_geom.h refactor was accepted without a GPU).
set -u
OUT=$1; shift
mkdir -p "$OUT"
L=/opt/rocm/lib/llvm/bin
cd "$(dirname "$0")/../../libavif_amd/csrc"
[ $# -eq 0 ] && set -- tile_*.o tilefx_*.o
for o in "$@"; do
  $L/llvm-objcopy --dump-section .hip_fatbin="$OUT/$o.fb" "$o" 2>/dev/null || continue
  $L/clang-offload-bundler --unbundle --type=o --input="$OUT/$o.fb" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$OUT/$o.co" 2>/dev/null
  $L/llvm-objdump -d --no-show-raw-insn "$OUT/$o.co" | sed -e 's/^ *[0-9a-f]*:\?//' -e 's_ *//.*$__' -e '/file format/d' > "$OUT/$o.s"
  rm -f "$OUT/$o.fb" "$OUT/$o.co"
done
cat "$OUT"/*.s | md5sum
