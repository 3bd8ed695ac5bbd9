# the -m gpu suite, with the parity prints (gpurun_out/${TAG}_gputest.log)
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/${TAG}_gputest.log 2>&1; tail -3 gpurun_out/${TAG}_gputest.log
grep -E "FAILED|Error|assert" gpurun_out/${TAG}_gputest.log | head -40
