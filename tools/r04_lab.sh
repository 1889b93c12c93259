#!/bin/bash
# lookup lab on the GPU box: HBM-resident tables, the configs[1] tables, and the giants-confined control
set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_lab; mkdir -p $out
timeout -k 10 300 tools/probe/lookup_lab 100000000 10000000 8192 64 0 > $out/lab_100m.txt 2>&1
echo "100m done"; cat $out/lab_100m.txt
timeout -k 10 300 tools/probe/lookup_lab 1000000 1000000 8192 64 0 > $out/lab_1m.txt 2>&1
echo "1m done"; cat $out/lab_1m.txt
timeout -k 10 300 tools/probe/lookup_lab 100000000 10000000 8192 64 1 > $out/lab_100m_small_giants.txt 2>&1
echo "control done"; cat $out/lab_100m_small_giants.txt
