set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_dgemm_gpu.py -x -q -m gpu 2>&1 | tail -30
