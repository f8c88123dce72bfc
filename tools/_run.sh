set -e
mkdir -p gpurun_out/r05e
python bench.py > gpurun_out/r05e/bench.json 2> gpurun_out/r05e/bench.err
echo bench done
bash tools/profile_config.sh c4 r05e_c4 > gpurun_out/r05e/c4.log 2>&1
python bench.py --config c4 > gpurun_out/r05e_c4/bench.json 2> gpurun_out/r05e_c4/bench.err
echo c4 done
bash tools/profile_config.sh c5 r05e_c5 > gpurun_out/r05e/c5.log 2>&1
python bench.py --config c5 > gpurun_out/r05e_c5/bench.json 2> gpurun_out/r05e_c5/bench.err
echo c5 done
python bench.py --train-step eval > gpurun_out/r05e/train_step_eval.json 2> gpurun_out/r05e/train_eval.err
python bench.py --train-step train > gpurun_out/r05e/train_step_train.json 2> gpurun_out/r05e/train_train.err
echo train done
python tools/schedule_waits.py > gpurun_out/r05e/schedule_waits.txt 2>&1
python tools/ablate.py > gpurun_out/r05e/ablation.txt 2>&1
echo all done
