set -e
mkdir -p gpurun_out/r5aq
python -m pytest tests/test_grad_gpu.py -x -q > gpurun_out/r5aq/tests.txt 2>&1
python tools/train_step_time.py 8 8192 train 10 > gpurun_out/r5aq/step_train.txt 2>&1
python tools/train_step_time.py 8 8192 eval 10 > gpurun_out/r5aq/step_eval.txt 2>&1
python tools/train_breakdown.py 8 8192 eval > gpurun_out/r5aq/bd_eval.txt 2>&1
