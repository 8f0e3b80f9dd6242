#!/bin/bash
# shader clock and power (rocm-smi) while one kind of load runs: usage tools/smi_sample.sh MODE (1: MFMA from registers, 2: GEMM)
tools/sclk_probe $1 5 &
LP=$!
sleep 1.0
for i in 1 2 3 4 5; do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1 | sed 's/.*(\(.*\)).*/sclk \1/' | tr '\n' ' '
  rocm-smi --showpower 2>/dev/null | grep -i "power (w)\|Average\|Current Socket" | head -1 | sed 's/.*: *//' 
  sleep 0.6
done
wait $LP
