#!/bin/bash
# Runs the GPU suite on the box the way every gpurun call of it should: ONE pytest process, its complete output (stdout + stderr) kept in a file
# under gpurun_out/ that is merged back -- never behind `| tail` --, the Python fault handler on (a fatal signal inside a native call leaves a
# traceback of every thread), and the id of every test written to gpurun_out/<tag>_trace.txt before its body runs (tests/conftest.py), so that a
# run that dies names the test it died in.
#   gpurun --timeout 1200 -- 'bash tools/gpu_suite.sh r4a [extra pytest args]'
set -o pipefail
TAG=${1:-suite}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/${TAG}_pytest.txt
export BBGPU_TEST_TRACE=$ROOT/gpurun_out/${TAG}_trace.txt
: > $BBGPU_TEST_TRACE
cd $ROOT
python -X faulthandler -m pytest tests -m gpu -x -q -p faulthandler -o faulthandler_timeout=600 "$@" > $OUT 2>&1
RC=$?
echo "pytest exit code $RC" >> $OUT
tail -n 15 $OUT
exit $RC
