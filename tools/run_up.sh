python -m pytest tests/test_encoder_gpu.py tests/test_train_gpu.py tests/test_unet3d_gpu.py -q -x 2>&1 | tail -4
python3 tools/bench_extra.py train 2>&1 | grep -v "amdgpu.ids" | tail -3 | cut -c1-600
