# end-of-round evidence on the last sources: full GPU suite, encoder timeline, training kernel table, fusion PMC, decode PMC + bench
cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r04_gputest_tail.txt
bash tools/probe/enc_tl.sh r04 > /dev/null 2>&1; tail -1 gpurun_out/enc_timeline_r04.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train_r04 -o train -- python3 /root/repo/tools/train_hip_prof.py 10 > /root/repo/gpurun_out/r04_train_hip.txt 2>&1; find /root/repo/gpurun_out/prof_train_r04 -name "*kernel_trace*" -delete)
tail -1 gpurun_out/r04_train_hip.txt
TAG=r04 bash tools/pmc_fusion.sh > gpurun_out/r04_pmc_fusion.log 2>&1; tail -3 gpurun_out/r04_pmc_fusion.log | cut -c1-200
TAG=r04 bash tools/collect_profiles.sh > gpurun_out/r04_collect.log 2>&1; tail -4 gpurun_out/r04_collect.log
cp gpurun_out/prof_r04/pmc_summary.csv profiles/r04_pmc_summary.csv
timeout 300 python3 bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"; head -c 400 gpurun_out/r04_bench_final.json
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
