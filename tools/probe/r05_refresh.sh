# after the last kernel changes of the round (MANO slices, mean pooling): the parts of r05_final.sh they touch
cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r05_gputest_tail.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train_r05 -o train -- python3 /root/repo/tools/train_hip_prof.py 10 > /root/repo/gpurun_out/r05_train_hip.txt 2>&1; find /root/repo/gpurun_out/prof_train_r05 -name "*kernel_trace*" -delete)
tail -1 gpurun_out/r05_train_hip.txt
timeout 400 python3 bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; echo "bench rc=$?"
timeout 900 python3 tools/bench_extra.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r05_bench_extra.jsonl; wc -l gpurun_out/r05_bench_extra.jsonl
python3 tools/probe/train_hip_step.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | head -40 > gpurun_out/r05_train_hip_step.txt; head -2 gpurun_out/r05_train_hip_step.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
bash tools/probe/enc_tl.sh r05 > /dev/null 2>&1; tail -1 gpurun_out/enc_timeline_r05.txt
