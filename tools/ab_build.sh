#!/bin/bash
# Build a variant with extra compile flags and FAIL LOUDLY if the build fails (a stale .so silently invalidates an A/B).
# usage: tools/ab_build.sh "<flags>"
DN_EXTRA_FLAGS="$1" python -m diffnet_amd.build --force > /tmp/ab_build.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then echo "BUILD FAILED for [$1]"; grep -iE "error" /tmp/ab_build.log | head -5; exit 1; fi
echo "built [$1]"
