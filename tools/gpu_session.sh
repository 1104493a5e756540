set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_fullsize.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/dgelu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/dgelu_tests.log | cut -c1-300
[ $rc -eq 0 ] || { tail -40 gpurun_out/dgelu_tests.log | cut -c1-200; exit $rc; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_dgelu
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dgelu/t1024 -- python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_dgelu_1024.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dgelu/t128 -- python bench.py --mode train --batch 128 --steps 25 --warmup 5 --no-cpu-baseline > gpurun_out/prof_dgelu_128.log 2>&1 || exit 1
find gpurun_out/prof_dgelu -name "*kernel_trace.csv" -delete
for d in t1024 t128; do f=$(find gpurun_out/prof_dgelu/$d -name "*kernel_stats.csv" | head -1); echo $d; grep -E "gemm256_kernel<unsigned short, (5|4), 8>|gemm128_kernel<5>|gemm128_kernel<4>" $f | cut -c1-200; done
tail -1 gpurun_out/prof_dgelu_1024.log | cut -c1-200
