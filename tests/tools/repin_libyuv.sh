#!/bin/bash
# repin_libyuv.sh <libyuv checkout> [reference tree] -- re-pins the integer-path oracle (oracle/libyuv_oracle.c) and the golden fixtures
# (tests/golden/yuvlib_*.npz) against the libyuv the reference ACTUALLY pins: chromium.googlesource.com/libyuv/libyuv @ 5d03bf9
# (LIBYUV_VERSION 1949; /root/reference/cmake/Modules/LocalLibyuv.cmake:4, ext/libyuv.cmd:19).
#
# Today the oracle is pinned against the only libyuv-enabled libavif binary available offline: Pillow's bundled libavif 1.4.1 + libyuv 1922.
# The skew 1922 -> 1949 cannot be quantified without libyuv's source, and this container has no network (VERDICT r04 missing #3).  On the first
# box that has a checkout, this is the whole re-pin:
#
#     git clone https://chromium.googlesource.com/libyuv/libyuv && git -C libyuv checkout 5d03bf9
#     bash tests/tools/repin_libyuv.sh $PWD/libyuv            # builds, checks, prints what (if anything) differs
#     bash tests/tools/repin_libyuv.sh $PWD/libyuv --write    # ... and regenerates tests/golden/yuvlib_*.npz from the 1949 build
#
# What it does -- no cmake, the same recipe style as oracle/Makefile (plain gcc / g++ on the sources where they lie):
#   1. compiles libyuv's source/*.cc (portable C row functions only: -DLIBYUV_DISABLE_X86 -DLIBYUV_DISABLE_NEON, i.e. the arithmetic the
#      SIMD paths are bit-exact with by libyuv's own unit tests) into oracle/_ref/libyuv1949/*.o;
#   2. compiles the reference's src/*.c with -DAVIF_LIBYUV_ENABLED=1 against libyuv's headers and links both into
#      oracle/_ref/libavif_yuvlib1949.so (a libavif built WITH the pinned libyuv, no codecs);
#   3. runs tests/test_libyuv_oracle.py with AVIFHIP_LIBYUV_BINARY pointing at it: every comparison that today runs against Pillow's
#      1922 binary (8 000+ configurations, all 2^24 RGB triples through the luma formulas, all 65 536 attenuate pairs) then runs against 1949.
#      Green = the restatement needs no change; red = the failing cases name the row function whose arithmetic moved;
#   4. with --write: tests/tools/make_golden_libyuv.py regenerates the fixtures from that binary (commit them with the version in the message).
# NOT RUN HERE: no libyuv checkout exists offline (the reference vendors only libyuv's scaler).  Steps 2-4 mirror recipes that do run in this
# container (oracle/Makefile builds the reference the same way without libyuv; test_libyuv_oracle.py runs against Pillow's binary through
# the same AVIFHIP_LIBYUV_BINARY switch in tests/oracle_lib.py).
set -eu
YUV=${1:?usage: repin_libyuv.sh <libyuv checkout> [--write] [reference tree]}
shift
WRITE=0; REF=/root/reference
for a in "$@"; do if [ "$a" = "--write" ]; then WRITE=1; else REF=$a; fi; done
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/oracle/_ref
[ -f "$YUV/include/libyuv/version.h" ] || { echo "repin_libyuv: $YUV is not a libyuv checkout (include/libyuv/version.h missing)" >&2; exit 2; }
[ -d "$REF/src" ] || { echo "repin_libyuv: reference tree $REF not found" >&2; exit 2; }
VERSION=$(sed -n 's/^#define LIBYUV_VERSION \([0-9]*\).*/\1/p' "$YUV/include/libyuv/version.h")
echo "repin_libyuv: libyuv $VERSION from $YUV (the reference pins 1949), reference $REF"
mkdir -p "$OUT/libyuv$VERSION"
for f in "$YUV"/source/*.cc; do
  g++ -O2 -fPIC -w -DLIBYUV_DISABLE_X86 -DLIBYUV_DISABLE_NEON -DLIBYUV_DISABLE_SVE -DLIBYUV_DISABLE_SME -I"$YUV/include" -c "$f" -o "$OUT/libyuv$VERSION/$(basename "$f" .cc).o"
done
SRCS="alpha avif colr colrconvert diag exif gainmap io mem obu properties rawdata read reformat reformat_libsharpyuv reformat_libyuv sampletransform scale stream utils write"
mkdir -p "$OUT/objy"
for s in $SRCS; do
  gcc -O3 -DNDEBUG -std=gnu11 -fPIC -w -DAVIF_LIBYUV_ENABLED=1 -I"$REF/include" -I"$YUV/include" -c "$REF/src/$s.c" -o "$OUT/objy/$s.o"
done
g++ -shared -o "$OUT/libavif_yuvlib$VERSION.so" "$OUT"/objy/*.o "$OUT/libyuv$VERSION"/*.o -lm -lpthread -Wl,--no-undefined
rm -rf "$OUT/objy"
echo "repin_libyuv: built $OUT/libavif_yuvlib$VERSION.so"
cd "$ROOT"
AVIFHIP_LIBYUV_BINARY="$OUT/libavif_yuvlib$VERSION.so" python -m pytest tests/test_libyuv_oracle.py -q -x && echo "repin_libyuv: oracle/libyuv_oracle.c agrees with libyuv $VERSION on every pinned comparison"
if [ $WRITE = 1 ]; then
  AVIFHIP_LIBYUV_BINARY="$OUT/libavif_yuvlib$VERSION.so" python tests/tools/make_golden_libyuv.py
  python -m pytest tests/test_golden.py -q && echo "repin_libyuv: tests/golden/yuvlib_*.npz regenerated from libyuv $VERSION -- commit them"
fi
