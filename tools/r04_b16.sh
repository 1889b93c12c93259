#!/bin/bash
# per-graph-launch overhead of the HIP runtime under its documented-by-name switches (probe section 4 only matters)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b16; mkdir -p $out
run() { name=$1; shift; env "$@" timeout -k 10 120 tools/probe/graph_setparams 50 > $out/$name.txt 2>&1; echo "== $name ($*)"; grep -A8 "^4\." $out/$name.txt | tail -7; }
run default X=1
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run kernarg_copy_opt_0 DEBUG_HIP_KERNARG_COPY_OPT=0
run kernarg_copy_opt_1 DEBUG_HIP_KERNARG_COPY_OPT=1
run graph_batch_1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graph_batch_256 DEBUG_HIP_GRAPH_BATCH_SIZE=256
run cpu_wait_0 ROC_CPU_WAIT_FOR_SIGNAL=0
run force_graph_queues_1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run hdp_wa_0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
