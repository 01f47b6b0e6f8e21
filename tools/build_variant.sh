#!/bin/bash
# usage: tools/build_variant.sh <git-rev|WORK> <out.so>   -- builds libnesr_hip from a revision's csrc into build/
set -e
REV=$1; OUT=$2
D=$(mktemp -d)
mkdir -p $D/neural_enhanced_super_resolution_amd/csrc $D/include $(dirname $OUT)
if [ "$REV" = "WORK" ]; then
  cp neural_enhanced_super_resolution_amd/csrc/* $D/neural_enhanced_super_resolution_amd/csrc/; cp include/nesr_hip.h $D/include/
else
  for f in $(git ls-tree --name-only $REV neural_enhanced_super_resolution_amd/csrc/); do git show $REV:$f > $D/$f; done
  git show $REV:include/nesr_hip.h > $D/include/nesr_hip.h
fi
cd $D/neural_enhanced_super_resolution_amd/csrc
SRCS=""; for f in *.hip *.cpp; do SRCS="$SRCS -x hip $f"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I . -o $OLDPWD/$OUT $SRCS 2>&1 | grep -v "warning\|^$" || true
cd $OLDPWD; rm -rf $D; ls -la $OUT
