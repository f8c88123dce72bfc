set -e
mkdir -p gpurun_out/r5aj
python -m pytest tests/test_grad_gpu.py -x -q > gpurun_out/r5aj/tests.txt 2>&1
python tools/train_step_time.py 8 8192 eval > gpurun_out/r5aj/step_eval.txt 2>&1
python tools/train_step_time.py 8 8192 train > gpurun_out/r5aj/step_train.txt 2>&1
python tools/train_step_time.py 8 8192 eval >> gpurun_out/r5aj/step_eval.txt 2>&1
