#!/bin/bash
# bash tools/r04_tests.sh <log name> <pytest args...>   -- one pytest process on the GPU box, log under gpurun_out/r04/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
LOG=gpurun_out/r04/$1.log; shift
timeout -k 10 ${TEST_TIMEOUT:-800} python -m pytest "$@" > $LOG 2>&1; rc=$?
echo "tests rc=$rc"; tail -15 $LOG
exit $rc
