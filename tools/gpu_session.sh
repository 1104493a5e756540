set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q -m gpu --timeout 200 --timeout-method thread -s tests/test_hip_fullsize.py::test_cartpole_training_gradients_vs_oracle tests/test_hip_parity.py::test_seeded_loop_equals_injected_philox_noise 2>&1 | grep -E "^\[|passed|failed|^E " | head -20
