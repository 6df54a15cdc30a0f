#!/bin/bash
# Builds the library of the committed HEAD into decodingustools_amd/lib/libcallable_hip_base.so (for
# tools/ab_bench.sh), then rebuilds the working tree's.
set -e
cd "$(dirname "$0")/.."
git stash -q
trap 'git stash pop -q' EXIT
python -c "import decodingustools_amd.build as b; b.build()"
cp decodingustools_amd/lib/libcallable_hip.so decodingustools_amd/lib/libcallable_hip_base.so
trap - EXIT
git stash pop -q
python -c "import decodingustools_amd.build as b; b.build()"
echo built base + working tree
