#!/bin/bash
# Developer tool (container): the working tree's library with the in-kernel per-pass clocks (NHW_PROFILE) -> tools/dev/prof.so, for tools/dev/gpu_pass_profile.py on the GPU box
# (there: cp tools/dev/prof.so nhwcodec_amd/libnhwhip.so before, and the tree's own library back after).
set -e
D=$(mktemp -d)
cp -r nhwcodec_amd include "$D"/
rm -f "$D"/nhwcodec_amd/csrc/*.o "$D"/nhwcodec_amd/libnhwhip.so
(cd "$D" && NHW_PROFILE=1 python -c "from nhwcodec_amd.build import build; build(force=True)" > /dev/null)
cp "$D/nhwcodec_amd/libnhwhip.so" "$(dirname "$0")/prof.so"
rm -rf "$D"
echo "tools/dev/prof.so built"
