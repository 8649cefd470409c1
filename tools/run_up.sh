set -x
python -m pytest tests/test_encoder_gpu.py -q -x -k "skips_the_empty or block_flags" 2>&1 | tail -5
python -m pytest tests/test_unet3d_gpu.py -q -x -k "tile_flags" 2>&1 | tail -5
python -m pytest tests/test_config2_shipped_gpu.py -q -x 2>&1 | tail -5
bash tools/probe/enc_tl.sh 2>&1 | tail -40
