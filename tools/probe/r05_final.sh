# end-of-round evidence on the last sources: full GPU suite, encoder timeline, conv stamps, training kernel table, conv / fusion / decode PMC, bench
cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r05_gputest_tail.txt
bash tools/probe/enc_tl.sh r05 > /dev/null 2>&1; tail -1 gpurun_out/enc_timeline_r05.txt
bash tools/probe/up_stamps.sh > /dev/null 2>&1; cp gpurun_out/up_stamps.txt gpurun_out/r05_up_stamps.txt
(VTACO_HIP_LIB=/root/repo/variants/lib_hb.so timeout 300 python3 tools/diag_conv.py 64 32 32 2>&1 | grep -v amdgpu > gpurun_out/r05_conv_prologue_stamps.txt)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train_r05 -o train -- python3 /root/repo/tools/train_hip_prof.py 10 > /root/repo/gpurun_out/r05_train_hip.txt 2>&1; find /root/repo/gpurun_out/prof_train_r05 -name "*kernel_trace*" -delete)
tail -1 gpurun_out/r05_train_hip.txt
bash tools/pmc_conv.sh > gpurun_out/r05_pmc_conv.log 2>&1; tail -2 gpurun_out/r05_pmc_conv.log
TAG=r05 bash tools/pmc_fusion.sh > gpurun_out/r05_pmc_fusion.log 2>&1; tail -3 gpurun_out/r05_pmc_fusion.log | cut -c1-200
TAG=r05 bash tools/collect_profiles.sh > gpurun_out/r05_collect.log 2>&1; tail -4 gpurun_out/r05_collect.log
timeout 300 python3 bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; echo "bench rc=$?"; head -c 300 gpurun_out/r05_bench_final.json
timeout 900 python3 tools/bench_extra.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r05_bench_extra.jsonl; wc -l gpurun_out/r05_bench_extra.jsonl
python3 tools/probe/train_hip_step.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | head -40 > gpurun_out/r05_train_hip_step.txt; head -2 gpurun_out/r05_train_hip_step.txt
(hipcc --offload-arch=gfx950 -O3 -w -o /tmp/exp_probe tools/probe/exp_probe.hip && /tmp/exp_probe > gpurun_out/r05_exp_probe.txt)
(VTACO_HIP_LIB=/root/repo/variants/lib_wp.so VTACO_WP_PRINT=1 python3 tools/probe/wp_stamps.py 2>&1 | grep -A11 "second launch" > gpurun_out/r05_wp_stamps.txt)
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
