# rocprof kernel stats at the Cartpole shapes (18 tokens, 12 heads of 128): sampling B = 320 / 64 and a training step B = 128
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_cart
rm -rf $O && mkdir -p $O
X="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s320 -- python bench.py --case cartpole --steps 50 --warmup 5 $X > $O/s320.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s64 -- python bench.py --case cartpole --batch 64 --steps 50 --warmup 5 $X > $O/s64.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t128 -- python bench.py --case cartpole --mode train --batch 128 --augment --steps 20 --warmup 4 --no-cpu-baseline --no-roofline > $O/t128.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python tools/prof_summary.py $O/s320 55 10; python tools/prof_summary.py $O/s64 55 10; python tools/prof_summary.py $O/t128 24 16
tail -c 300 $O/t128.log
