cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/sf1 -o sf -- python /root/repo/tools/single_frame_profile.py 40 > /root/repo/gpurun_out/sf1.log 2>&1
tail -1 /root/repo/gpurun_out/sf1.log
