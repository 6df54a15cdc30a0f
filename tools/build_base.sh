#!/bin/bash
# Builds the library of the committed HEAD into decodingustools_amd/lib/libcallable_hip_base.so (for
# tools/ab_bench.sh) from a throwaway worktree -- the working tree and the stash are never touched --, then makes
# sure the working tree's own library is current.
set -e
cd "$(dirname "$0")/.."
root=$(pwd)
tmp=$(mktemp -d)
trap 'git worktree remove --force "$tmp/head" >/dev/null 2>&1 || true; rm -rf "$tmp"' EXIT
git worktree add --detach -q "$tmp/head" HEAD
(cd "$tmp/head" && python -c "import decodingustools_amd.build as b; b.build()")
cp "$tmp/head/decodingustools_amd/lib/libcallable_hip.so" "$root/decodingustools_amd/lib/libcallable_hip_base.so"
python -c "import decodingustools_amd.build as b; b.build()"
echo built base + working tree
