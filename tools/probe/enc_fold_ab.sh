export TMPDIR=/tmp; mkdir -p gpurun_out/et
python -m pytest tests/test_unet3d_gpu.py -x -q -m gpu 2>&1 | tail -4
for f in 0 1; do
  export VTACO_GN_FOLD=$f
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/et$f -o e -- python3 tools/enc_timeline.py > gpurun_out/et/run$f.log 2>&1
  python3 tools/enc_timeline.py gpurun_out/et$f/e_kernel_trace.csv > gpurun_out/et/timeline_fold$f.txt
done
paste -d'|' <(cut -c1-46,60-75 gpurun_out/et/timeline_fold1.txt) /dev/null | tail -30
tail -1 gpurun_out/et/timeline_fold0.txt
python3 tools/enc_prof.py; VTACO_GN_FOLD=0 python3 tools/enc_prof.py
