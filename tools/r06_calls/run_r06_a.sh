#!/bin/bash
# call a: first run of the packet walk (k_trace_packet): its tests, then the A/B table
set -x
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_packet.py -x -q > $O/test_packet.log 2>&1; echo "rc $?" >> $O/test_packet.log
tail -5 $O/test_packet.log
timeout 1500 python tools/ab_packet.py interior_primary interior atrium_primary atrium headline_primary headline > $O/ab_packet.log 2>&1; echo "rc $?" >> $O/ab_packet.log
cat $O/ab_packet.log | tail -20
