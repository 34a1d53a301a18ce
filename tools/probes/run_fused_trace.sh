# kernel trace of tools/nms_fused_ab.py: durations of the mask, sweep and fused kernels.  usage: run_fused_trace.sh <tag> [ab args]
tag=$1; shift
mkdir -p gpurun_out/r3k && cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3k/prof_$tag -o t -- python3 /root/repo/tools/nms_fused_ab.py --iters 10 "$@" > /root/repo/gpurun_out/r3k/prof_$tag.log 2>&1 || exit 1
grep nms_fused /root/repo/gpurun_out/r3k/prof_$tag.log | cut -c1-200
