# in-step duration of the stand-alone NMS kernels (two launches): bench under rocprofv3 with nms_fused=0.  usage: <tag>
mkdir -p gpurun_out/r3k && cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3k/prof_$1 -o b -- python3 /root/repo/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-miopen-benchmark --tuning nms_fused=0 > /root/repo/gpurun_out/r3k/prof_$1.log 2>&1
grep -h "nms_sweep\|nms_mask" /root/repo/gpurun_out/r3k/prof_$1/b_kernel_stats.csv | cut -d, -f1-7 | cut -c1-160
