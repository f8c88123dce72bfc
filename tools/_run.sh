set -e
mkdir -p gpurun_out/r5am
python -m pytest tests/test_grad_gpu.py -x -q > gpurun_out/r5am/tests.txt 2>&1
python tools/train_step_time.py 8 8192 train 10 > gpurun_out/r5am/step_train.txt 2>&1
python tools/train_step_time.py 8 8192 eval 10 > gpurun_out/r5am/step_eval.txt 2>&1
