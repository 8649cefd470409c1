# node timeline of one replay of the visual scene graph (kernels + copies): bash tools/probe/scene_graph_tl.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/sg_tl -o s -- python3 /root/repo/tools/probe/scene_graph_tl.py > /tmp/sg_run.txt 2>&1; tail -5 /tmp/sg_run.txt; find /tmp/sg_tl -name "*.csv" | head
ls /tmp/sg_tl/*/ 2>/dev/null | head; 
python3 /root/repo/tools/probe/scene_graph_tl.py $(find /tmp/sg_tl -name "*kernel_trace.csv") $(find /tmp/sg_tl -name "*memory_copy_trace.csv") > /root/repo/gpurun_out/scene_graph_tl.txt 2>&1
cat /root/repo/gpurun_out/scene_graph_tl.txt
