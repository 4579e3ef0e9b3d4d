#!/bin/bash
# GPU box: chol(P) beside the features - mode 3 (inside the fused launch) against mode 4 (k_chol2 on the side stream)
mkdir -p gpurun_out
for feats in 1900 500 2000; do
for mode in 3 4; do
  echo "== OVP_OVERLAP_MODE=$mode"
  OVP_OVERLAP_MODE=$mode timeout 120 python tools/k1_ab.py --time --time-feats $feats 2>&1 | tail -3
done
done > gpurun_out/mode4_ab.txt 2>&1
cat gpurun_out/mode4_ab.txt
