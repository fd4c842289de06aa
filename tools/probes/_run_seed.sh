python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "seeding or mean_shift" 2>&1 | tail -5
python tools/probes/ms_cfg5_time.py 2>&1 | grep -v amdgpu.ids
