# Re-collection of the two training-step kernel summaries of round 1 (run from the repo root through gpurun).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_final
mkdir -p $O && rm -rf $O/train128 $O/train1024
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train128 -- python bench.py --mode train --batch 128 --steps 25 --warmup 5 --no-cpu-baseline > $O/train128.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train1024 -- python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $O/train1024.log 2>&1
find $O -name "*kernel_trace.csv" -delete
ls -R $O/train128 $O/train1024
