#!/bin/bash
# usage: tools/build_flags.sh <out.so> [-DFLAG=V ...]   -- builds libnesr_hip from the work tree with extra compiler flags
set -e
OUT=$1; shift
mkdir -p $(dirname $OUT)
cd neural_enhanced_super_resolution_amd/csrc
SRCS=""; for f in *.hip *.cpp; do SRCS="$SRCS -x hip $f"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I . "$@" -o $OLDPWD/$OUT $SRCS 2>&1 | grep -E "error" || true
cd $OLDPWD; ls -la $OUT
