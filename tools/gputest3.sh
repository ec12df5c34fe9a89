cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -15
