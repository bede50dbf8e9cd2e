for f in tests/test_gpu_core.py tests/test_gpu_mel.py tests/test_gpu_models.py tests/test_gpu_streaming.py tests/test_gpu_training.py tests/test_gpu_dataset.py; do
  echo "== $f" >> gpurun_out/split.log
  timeout -k 5 90 python -m pytest $f -x -q >> gpurun_out/split.log 2>&1
  echo "rc=$?" >> gpurun_out/split.log
  tail -3 gpurun_out/split.log
done
