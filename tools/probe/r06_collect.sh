# Round-6 evidence on the GPU box (every file goes into profiles/ only through tools/keep_evidence.py: empty files and tracebacks are
# refused).  Usage (GPU box): bash tools/probe/r06_collect.sh [decode|conv|fusion|wide|train|rest ...]   (default: all)
cd /root/repo; mkdir -p gpurun_out/r06
K="python3 tools/keep_evidence.py"
WANT=${*:-decode conv fusion wide train rest}
has(){ case " $WANT " in *" $1 "*) return 0;; esac; return 1; }
if has decode; then
  TAG=r06 bash tools/collect_profiles.sh > gpurun_out/r06_collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/r06_collect.log
  O=gpurun_out/prof_r06
  $K $O/pmc_summary.csv gpurun_out/r06/r06_pmc_summary.csv --must-contain decode_sources= --must-contain FETCH_SIZE
  $K "$(find $O/stats_decode -name '*kernel_stats.csv' | head -1)" gpurun_out/r06/r06_decode_kernel_stats.csv --must-contain decode_fwd
  $K "$(find $O/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/r06/r06_bench_kernel_stats.csv --must-contain decode_fwd
  $K $O/bench.json gpurun_out/r06/r06_bench.json --must-contain roofline
  $K $O/bench_extra.jsonl gpurun_out/r06/r06_bench_extra.jsonl --must-contain workload
fi
if has conv; then
  bash tools/pmc_conv.sh > gpurun_out/r06_pmc_conv.log 2>&1; echo "pmc_conv rc=$?"
  $K gpurun_out/pmc_conv/summary.csv gpurun_out/r06/r06_pmc_conv_summary.csv --must-contain conv_up8
  bash tools/probe/enc_tl.sh r06 > /dev/null 2>&1
  $K gpurun_out/enc_timeline_r06.txt gpurun_out/r06/r06_encoder_timeline.txt --must-contain "kernel time"
fi
if has fusion; then
  TAG=r06 bash tools/pmc_fusion.sh > gpurun_out/r06_pmc_fusion.log 2>&1; echo "pmc_fusion rc=$?"
  $K gpurun_out/prof_fusion_r06/pmc_summary.csv gpurun_out/r06/r06_fusion_pmc_summary.csv --must-contain fusion_attend
  $K "$(find gpurun_out/prof_fusion_r06/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/r06/r06_fusion_kernel_stats.csv --must-contain fusion_attend
fi
if has wide; then
  TAG=r06 bash tools/pmc_wide.sh > gpurun_out/r06_pmc_wide.log 2>&1; echo "pmc_wide rc=$?"
  $K gpurun_out/prof_wide_r06/pmc_summary.csv gpurun_out/r06/r06_pmc_wide_summary.csv --must-contain wide_sources= --must-contain FETCH_SIZE
  $K "$(find gpurun_out/prof_wide_r06/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/r06/r06_wide_kernel_stats.csv --must-contain decode_wide
fi
if has train; then
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train_r06 -o train -- python3 /root/repo/tools/train_hip_prof.py 10 > /root/repo/gpurun_out/r06_train_hip.txt 2>&1; find /root/repo/gpurun_out/prof_train_r06 -name "*kernel_trace*" -delete)
  $K "$(find gpurun_out/prof_train_r06 -name '*kernel_stats.csv' | head -1)" gpurun_out/r06/r06_train_kernel_stats.csv --must-contain conv3d
  python3 tools/probe/train_hip_step.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | head -40 > gpurun_out/r06_train_hip_step.txt
  $K gpurun_out/r06_train_hip_step.txt gpurun_out/r06/r06_train_hip_step.txt --must-contain "ms per step"
  python3 tools/probe/train_hip_step.py full 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | head -40 > gpurun_out/r06_train_full_step.txt
  $K gpurun_out/r06_train_full_step.txt gpurun_out/r06/r06_train_full_step.txt --must-contain "ms per step"
fi
if has rest; then
  bash tools/probe/scene_graph_tl.sh > /dev/null 2>&1; $K gpurun_out/scene_graph_tl.txt gpurun_out/r06/r06_scene_graph_timeline.txt --must-contain span
  bash tools/probe/plane_unet_tl.sh 3 > /dev/null 2>&1; $K gpurun_out/plane_unet_tl_3.txt gpurun_out/r06/r06_plane_unet_timeline.txt --must-contain "kernel time"
  bash tools/probe/plane_unet_tl.sh 24 bwd > /dev/null 2>&1; $K gpurun_out/plane_unet_tl_24bwd.txt gpurun_out/r06/r06_plane_unet_train_timeline.txt --must-contain "kernel time"
  tools/probe/barrier_probe > gpurun_out/r06_barrier_probe.txt 2>&1; $K gpurun_out/r06_barrier_probe.txt gpurun_out/r06/r06_barrier_probe.txt --must-contain "per barrier"
  python3 tools/probe/plane_unet_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_plane_unet_time.txt; $K gpurun_out/r06_plane_unet_time.txt gpurun_out/r06/r06_plane_unet_time.txt --must-contain "U-Net depth"
  python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06_gputest_tail.txt; $K gpurun_out/r06_gputest_tail.txt gpurun_out/r06/r06_gputest_tail.txt --must-contain passed --min-bytes 20
fi
ls -la gpurun_out/r06
