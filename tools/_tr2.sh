mkdir -p gpurun_out/r5u
python -m pytest tests/test_grad_gpu.py -x -q -k "training or step" > gpurun_out/r5u/grad_tests.txt 2>&1
python tools/train_step_time.py 8 8192 eval > gpurun_out/r5u/step_eval.txt 2>&1
python tools/train_step_time.py 8 8192 train > gpurun_out/r5u/step_train.txt 2>&1
