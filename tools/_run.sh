mkdir -p gpurun_out/r03_l; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -x -m gpu > gpurun_out/r03_l/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03_l/pytest_gpu.log | tail -3; grep -E "^E |Error" gpurun_out/r03_l/pytest_gpu.log | head -20
