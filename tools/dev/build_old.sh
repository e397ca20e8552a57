#!/bin/bash
# Developer tool (container): build the library of a commit next to the working tree's, for tools/dev/ab.sh on the GPU box.
# usage: bash tools/dev/build_old.sh [commit]   -> tools/dev/old.so
set -e
C=${1:-HEAD}
D=$(mktemp -d)
git archive "$C" nhwcodec_amd include | tar -x -C "$D"
(cd "$D" && python -c "from nhwcodec_amd.build import build; build(force=True)" > /dev/null)
cp "$D/nhwcodec_amd/libnhwhip.so" "$(dirname "$0")/old.so"
rm -rf "$D"
echo "tools/dev/old.so = $C"
