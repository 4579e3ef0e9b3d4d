#!/bin/bash
# GPU box: the whole GPU suite (result into gpurun_out/TAG_pytest_gpu.txt), then the quick step times of configs 3 / 2 / 4.
tag=${1:-r06_x}
python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.txt 2>&1
grep -E 'passed|failed' gpurun_out/${tag}_pytest_gpu.txt | tail -1
grep -E '^FAILED' gpurun_out/${tag}_pytest_gpu.txt | head -20
tools/gpu_bench_quick.sh config3
tools/gpu_bench_quick.sh config2
tools/gpu_bench_quick.sh config4 20
