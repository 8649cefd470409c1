cd /root/repo
python -m pytest tests/test_decode_wide_gpu.py -x -q -m gpu 2>&1 | tail -3
python3 tools/bench_extra.py wide 2>/dev/null | tee gpurun_out/r03_wide.jsonl | cut -c1-200
python3 tools/probe/fusion_err.py 2>/dev/null | tee gpurun_out/r03_fusion_err.txt
timeout 300 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
