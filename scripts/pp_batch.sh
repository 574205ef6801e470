python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
