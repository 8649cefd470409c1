# fusion A/B: V' written by the column-sum pass (default) against the pass of its own (VTACO_FUSION_SCALEV=0); projection tiles per wave
for v in 0 1 0 1; do
  VTACO_FUSION_SCALEV=$v python3 tools/probe/fusion_probe.py 256 2>&1 | grep "ms per" | sed "s/^/scalev=$v /"
done
for t in 8 ""; do
  VTACO_PROJ_TILES=$t python3 tools/probe/fusion_probe.py 256 2>&1 | grep "ms per"
done
python3 -m pytest tests/test_fusion_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -2
