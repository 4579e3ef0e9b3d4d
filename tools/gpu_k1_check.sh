#!/bin/bash
# On the GPU box (from the repo root): the K1 A/B table (tools/k1_ab.py) and the tests that go through K1.  Output under gpurun_out/.
mkdir -p gpurun_out
timeout ${K1_AB_LIMIT:-200} python tools/k1_ab.py --time > gpurun_out/k1_ab.txt 2>&1; echo "k1_ab rc=$?" >> gpurun_out/k1_ab.txt
tail -40 gpurun_out/k1_ab.txt
if [ "$1" != "notests" ]; then
  timeout ${K1_TEST_LIMIT:-330} python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fresh or calibration or empty or full_size or config2_full or larger_states or round4_shortcuts or config3_whole or sharded or rccl" > gpurun_out/k1_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/k1_tests.txt
  tail -8 gpurun_out/k1_tests.txt
fi
