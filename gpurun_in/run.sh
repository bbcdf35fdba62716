timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -n 25 gpurun_out/t.log
