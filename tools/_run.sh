set -e
mkdir -p gpurun_out/r5al
python -m pytest tests/test_ops_gpu.py -x -q -k "scatter or reference_api" > gpurun_out/r5al/tests.txt 2>&1
python -m pytest tests/test_grad_gpu.py -x -q >> gpurun_out/r5al/tests.txt 2>&1
python tools/train_step_time.py 8 8192 train 10 > gpurun_out/r5al/step_train.txt 2>&1
python tools/train_step_time.py 8 8192 eval 10 > gpurun_out/r5al/step_eval.txt 2>&1
