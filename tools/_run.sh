set -e
mkdir -p gpurun_out/r5ap
export TMPDIR=/tmp
python -m pytest tests/test_grad_gpu.py -x -q > gpurun_out/r5ap/tests.txt 2>&1
python tools/train_step_time.py 8 8192 train 10 > gpurun_out/r5ap/step_train.txt 2>&1
python tools/train_step_time.py 8 8192 eval 10 > gpurun_out/r5ap/step_eval.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r5ap/trace -o train -- python tools/train_step_time.py 8 8192 eval > gpurun_out/r5ap/traced.txt 2>&1
python tools/rocpd_top.py $(find gpurun_out/r5ap/trace -name '*.db' | head -1) 30 > gpurun_out/r5ap/top_eval.txt 2>&1
rm -rf gpurun_out/r5ap/trace
