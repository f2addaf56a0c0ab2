#!/bin/bash
# TEST INFRASTRUCTURE (oracle/): puts the UNMODIFIED reference where bench.py's cpu_baseline leg can time it on the GPU box.
#
# The reference is a pure-Python package (nothing to compile): the "build" is a verbatim copy of /root/reference/neurodiffeq
# plus the two import shims the golden script uses (tests/golden/_refshim: `seaborn`, `ordered_set` -- plotting / container
# dependencies that are not installed and that no code on the timed path executes) into oracle/_ref/.  oracle/_ref/ is
# git-ignored (no reference source ever enters the history) but not gpurun-ignored, so it travels with the snapshot like
# the built .so files.  Only bench.py (cpu_baseline) and oracle/ref_bench.py ever import from it.
#   usage: oracle/make_ref.sh [/root/reference]
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${1:-/root/reference}
if [ ! -d "$REF/neurodiffeq" ]; then
  echo "oracle/make_ref.sh: no reference at $REF (GPU box: the prebuilt oracle/_ref travels with the snapshot)"; exit 0
fi
[ -d "$HERE/_ref" ] && chmod -R u+w "$HERE/_ref"
rm -rf "$HERE/_ref"
mkdir -p "$HERE/_ref"
cp -r "$REF/neurodiffeq" "$HERE/_ref/neurodiffeq"
chmod -R u+w "$HERE/_ref"
cp -r "$HERE/../tests/golden/_refshim/seaborn" "$HERE/../tests/golden/_refshim/ordered_set" "$HERE/_ref/"
find "$HERE/_ref" -name __pycache__ -type d -prune -exec rm -rf {} +
# revision: the reference's git commit, or -- a checkout without .git -- its declared version plus a digest of the package tree
REV=$(cd "$REF" && git rev-parse HEAD 2>/dev/null) || \
  REV="version $(sed -n "s/.*version *= *['\"]\([^'\"]*\)['\"].*/\1/p" "$REF/setup.py" 2>/dev/null | head -1), tree-sha256 $(cd "$REF/neurodiffeq" && find . -name '*.py' | LC_ALL=C sort | xargs sha256sum | sha256sum | cut -c1-16)"
echo "$REV" > "$HERE/_ref/REVISION"
echo "oracle/_ref: reference package at revision $(cat "$HERE/_ref/REVISION") ($(find "$HERE/_ref/neurodiffeq" -name '*.py' | wc -l) files)"
