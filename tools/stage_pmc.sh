#!/bin/bash
# usage (on the GPU box): tools/stage_pmc.sh {sc16|uc8} tag  -- VALU/cycle counters per stage of the scan kernel
fmt=$1; tag=$2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d gpurun_out/pmc_$tag --output-format csv -- python3 tools/stage_time.py $fmt > gpurun_out/pmc_$tag.log 2>&1
