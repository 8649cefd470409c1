# the round's remaining evidence in one GPU call: wide-decoder tests + bench line, conv / fusion counters, encoder timeline, bench with the committed PMC source
cd /root/repo
python -m pytest tests/test_decode_wide_gpu.py -x -q -m gpu 2>&1 | tail -3
python3 tools/bench_extra.py wide 2>/dev/null | tee gpurun_out/r03_wide.jsonl | cut -c1-200
bash tools/pmc_conv.sh 2>&1 | tail -3
TAG=r03 bash tools/pmc_fusion.sh 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/et_r03 -o e -- python3 /root/repo/tools/enc_timeline.py > /dev/null 2>&1
python3 /root/repo/tools/enc_timeline.py /root/repo/gpurun_out/et_r03/e_kernel_trace.csv > /root/repo/gpurun_out/r03_encoder_timeline.txt; tail -2 /root/repo/gpurun_out/r03_encoder_timeline.txt
rm -rf /root/repo/gpurun_out/et_r03
cd /root/repo && timeout 300 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
