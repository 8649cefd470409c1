# timeline of the plane U-Net's launches on the GPU box.  Usage: bash tools/probe/plane_unet_tl.sh [n_img] [bwd]
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o e -- python3 /root/repo/tools/probe/plane_unet_tl.py run ${1:-3} $2 > /dev/null 2>&1
python3 /root/repo/tools/probe/plane_unet_tl.py /tmp/pt/e_kernel_trace.csv | tee /root/repo/gpurun_out/plane_unet_tl_${1:-3}$2.txt
