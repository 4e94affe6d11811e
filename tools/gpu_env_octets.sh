#!/bin/bash
# fused env layer at small batches: eight lanes per env (rb_select_kernel(2)) against two lanes per env (5) and one env per lane (1); parity first
cd /root/repo
TAG=${1:-r5_a}
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_env_layer_gpu.py tests/test_env_golden_gpu.py tests/test_ppo.py tests/test_policy_gpu.py -x -q -m gpu > gpurun_out/$TAG/env_octets_tests.log 2>&1 || { tail -30 gpurun_out/$TAG/env_octets_tests.log; exit 1; }
tail -3 gpurun_out/$TAG/env_octets_tests.log
for integ in rk4 euler; do for k in 2 5 1; do
  VECENV_GRAPH=1 VECENV_KERNEL=$k VECENV_INTEGRATOR=$integ VECENV_SIZES=256,1024,2048,4096,8192,12288,16384,24576 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null || exit 1
done; done > gpurun_out/$TAG/env_octets_sweep.log
cat gpurun_out/$TAG/env_octets_sweep.log
