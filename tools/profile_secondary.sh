# rocprof kernel stats of the secondary bench legs (CIFAR-10 shapes B = 1280 bf16, mnist shapes fp32 B = 320): a look for kernel
# instances that only these shapes reach
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_sec
rm -rf $O && mkdir -p $O
X="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cifar -- python bench.py --case cifar --batch 1280 --steps 20 --warmup 3 $X > $O/cifar.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32 -- python bench.py --dtype fp32 --steps 20 --warmup 3 $X > $O/fp32.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python tools/prof_summary.py $O/cifar 23 14; python tools/prof_summary.py $O/fp32 23 14
