#!/bin/bash
# builds tools/probe/score_bwd_lab against the in-tree library and prints register / scratch use of every lab kernel
set -e
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Ijodalrob-twotower_amd/csrc -Itools/probe"
/opt/rocm/bin/hipcc $FL -S --cuda-device-only -o /tmp/lab.s tools/probe/score_bwd_lab.hip 2>/dev/null
grep "^_Z7lab_bwd.*:\|; NumVgprs\|ScratchSize" /tmp/lab.s | grep -v "^\s*\.\|amdhsa" | paste - - - | awk '{print $1, $4, $5, $6, $7, $8, $9}'
/opt/rocm/bin/hipcc $FL -o tools/probe/score_bwd_lab tools/probe/score_bwd_lab.hip -Ljodalrob-twotower_amd -ltwotower_hip -Wl,-rpath,'$ORIGIN/../../jodalrob-twotower_amd' 2>/dev/null
ls -la tools/probe/score_bwd_lab
