# end-of-round evidence on the last sources: full GPU suite, encoder timeline, default bench
cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r03_gputest_tail.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/et_r03 -o e -- python3 /root/repo/tools/enc_timeline.py > /dev/null 2>&1)
python3 tools/enc_timeline.py gpurun_out/et_r03/e_kernel_trace.csv > gpurun_out/r03_encoder_timeline.txt; tail -1 gpurun_out/r03_encoder_timeline.txt
rm -rf gpurun_out/et_r03
timeout 300 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
python3 tools/enc_prof.py
