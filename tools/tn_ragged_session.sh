set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_hip_training.py tests/test_hip_fullsize.py -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/tn_ragged_tests.log 2>&1 || { tail -30 gpurun_out/tn_ragged_tests.log; exit 1; }
tail -2 gpurun_out/tn_ragged_tests.log
bash tools/ab_train_env.sh "--case cifar_big --batch 68 --augment --steps 6 --warmup 2" X=0
bash tools/ab_train_env.sh "--batch 100 --steps 20 --warmup 4" X=0 GPT_WGRAD_TN=0
