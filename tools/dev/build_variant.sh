#!/bin/bash
# Developer tool (container): the working tree's library with extra compiler flags -> tools/dev/<name>.so   usage: bash tools/dev/build_variant.sh <name> <flags...>
set -e
NAME=$1; shift
D=$(mktemp -d)
cp -r nhwcodec_amd include "$D"/
rm -f "$D"/nhwcodec_amd/csrc/*.o "$D"/nhwcodec_amd/libnhwhip.so
(cd "$D" && NHW_EXTRA_FLAGS="$*" python -c "from nhwcodec_amd.build import build; build(force=True)" > /dev/null)
cp "$D/nhwcodec_amd/libnhwhip.so" "$(dirname "$0")/$NAME.so"
rm -rf "$D"
echo "tools/dev/$NAME.so built with $*"
