#!/bin/bash
# round 4, GPU call M: leaf_min with the pooled leaf phase (a phase now costs about one round trip whatever the number of pairs)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04m
( SWEEP_TAG=r04m SWEEP_OPT=LEAF_MIN:8,12,16,20,24,28,36 SWEEP_BATCHES=32 SWEEP_DEPTHS=2,5 timeout 1500 python tools/sweep_r03.py headline interior atrium 2>&1 | tail -44 ) > gpurun_out/r04m/leaf_min.txt
cat gpurun_out/r04m/leaf_min.txt
