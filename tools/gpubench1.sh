set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -20
