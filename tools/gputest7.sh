cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_staging_gpu.py -x -q -m gpu 2>&1 | tail -5
