cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_cg_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python examples/grm_solve_cg.py --snps 250000 --indiv 100000 --max-iter 30 2>&1 | tail -2
