# encoder timeline of the current build on the GPU box: kernel list of ONE encode + total kernel time.  Usage: bash tools/probe/enc_tl.sh [tag]
cd /tmp && export TMPDIR=/tmp
TAG=${1:-x}
rocprofv3 --kernel-trace --output-format csv -d /tmp/et_$TAG -o e -- python3 /root/repo/tools/enc_timeline.py > /dev/null 2>&1
python3 /root/repo/tools/enc_timeline.py /tmp/et_$TAG/e_kernel_trace.csv > /root/repo/gpurun_out/enc_timeline_$TAG.txt
grep -E "conv3d|kernel time" /root/repo/gpurun_out/enc_timeline_$TAG.txt
