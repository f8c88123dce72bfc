mkdir -p gpurun_out/r5d
python tools/pointconv_ab.py > gpurun_out/r5d/pc_new.txt 2>&1
MCP_HIP_LIB=$PWD/build_old_r4.so python tools/pointconv_ab.py > gpurun_out/r5d/pc_old.txt 2>&1
python tools/step_time.py > gpurun_out/r5d/step_new.txt 2>&1
MCP_HIP_LIB=$PWD/build_old_r4.so python tools/step_time.py > gpurun_out/r5d/step_old.txt 2>&1
python tools/step_time.py >> gpurun_out/r5d/step_new.txt 2>&1
python -m pytest tests -m gpu -x -q -k "knn_cosine or cosine or pointconv or forward or determinism or linear" > gpurun_out/r5d/tests.txt 2>&1
