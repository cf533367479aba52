#!/bin/bash
# call c: verification of the packet walk: the whole GPU suite (default options), the batch / sample / parity tests with the packet walk forced, fuzz seeds drawing it, A/B
set -x
O=gpurun_out/r06c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "rc $?" >> $O/suite.log
tail -3 $O/suite.log
IDKPT_PACKET=2 timeout 1500 python -m pytest tests/test_gpu_batching.py tests/test_gpu_samples.py tests/test_gpu_parity.py tests/test_gpu_defer.py tests/test_gpu_wide.py tests/test_gpu_fullsize.py -x -q > $O/suite_packet_forced.log 2>&1; echo "rc $?" >> $O/suite_packet_forced.log
tail -3 $O/suite_packet_forced.log
FUZZ_BLASES=1,1 timeout 1500 python tools/fuzz_parity.py 600 0 > $O/fuzz_one_blas_600.log 2>&1; echo "rc $?" >> $O/fuzz_one_blas_600.log
tail -2 $O/fuzz_one_blas_600.log
timeout 1200 python tools/fuzz_parity.py 300 20000 > $O/fuzz_300.log 2>&1; echo "rc $?" >> $O/fuzz_300.log
tail -2 $O/fuzz_300.log
AB_MODES=0,1,0,1 timeout 900 python tools/ab_packet.py interior interior_20_samples interior_d5 atrium headline headline_20_samples > $O/ab_packet_auto.log 2>&1; echo "rc $?" >> $O/ab_packet_auto.log
