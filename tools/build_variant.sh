#!/bin/bash
# Builds libpt_amd.so of a given git revision (or of the working tree: "WORK") into
# thu-acg-f2024-path-tracer_amd/variants/libpt_amd_<name>.so for in-session A/B runs (PT_AMD_LIB=...).
# Usage: tools/build_variant.sh <name> <rev|WORK> [extra make flags, e.g. EXTRA=-DPT_X=1]
set -e
NAME=$1; REV=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG="$ROOT/thu-acg-f2024-path-tracer_amd"
mkdir -p "$PKG/variants"
if [ "$REV" = "WORK" ]; then          # a clean build of the working tree (so that EXTRA=... flags take effect)
  TMP=$(mktemp -d)
  mkdir -p "$TMP/thu-acg-f2024-path-tracer_amd"
  cp -r "$PKG/csrc" "$PKG/host" "$PKG/Makefile" "$TMP/thu-acg-f2024-path-tracer_amd/"
  cp -r "$ROOT/include" "$TMP/"
  SRC="$TMP/thu-acg-f2024-path-tracer_amd"
  make -C "$SRC" -j4 libpt_amd.so "$@" > /dev/null
else
  TMP=$(mktemp -d)
  git -C "$ROOT" archive "$REV" thu-acg-f2024-path-tracer_amd include | tar -x -C "$TMP"
  SRC="$TMP/thu-acg-f2024-path-tracer_amd"
  make -C "$SRC" -j4 libpt_amd.so "$@" > /dev/null
fi
cp "$SRC/libpt_amd.so" "$PKG/variants/libpt_amd_$NAME.so"
echo "built variants/libpt_amd_$NAME.so from $REV"
