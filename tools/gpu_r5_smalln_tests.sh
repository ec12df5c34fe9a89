#!/bin/bash
# round 5: the guarded small-n routes with the device-side class verdict (no host wait), the fp64 kernel behind them, the async test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_async_gpu.py tests/test_small_n_gpu.py tests/test_i8_tn_gpu.py tests/test_single_orientation_gpu.py tests/test_edge_gpu.py tests/test_dgemm_gpu.py tests/test_cg_gpu.py tests/test_engine_i8_gpu.py tests/test_engine_i8_exact_gpu.py tests/test_property_gpu.py tests/test_workspace_gpu.py tests/test_host_pipeline_gpu.py tests/test_shard_gpu.py -x -q -m gpu > gpurun_out/r5_smalln_tests.log 2>&1
rc=$?
tail -40 gpurun_out/r5_smalln_tests.log
exit $rc
