cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_c_driver_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-700
