#!/bin/bash
# call b: the packet walk rewritten in the mask domain: tests, A/B table at 28 and 32 waves per CU
set -x
O=gpurun_out/r06b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_packet.py -x -q > $O/test_packet.log 2>&1; echo "rc $?" >> $O/test_packet.log
tail -5 $O/test_packet.log
AB_MODES=0,2,0,2 timeout 1500 python tools/ab_packet.py interior_primary interior atrium_primary atrium headline_primary headline interior_one_frame atrium_one_frame > $O/ab_packet_28.log 2>&1; echo "rc $?" >> $O/ab_packet_28.log
AB_MODES=2,2 PACKET_WAVES=32 timeout 1500 python tools/ab_packet.py interior_primary atrium_primary headline_primary > $O/ab_packet_32.log 2>&1; echo "rc $?" >> $O/ab_packet_32.log
AB_MODES=2,2 PACKET_WAVES=20 timeout 1500 python tools/ab_packet.py interior_primary atrium_primary headline_primary > $O/ab_packet_20.log 2>&1; echo "rc $?" >> $O/ab_packet_20.log
