# Round-3 profile collection on one MI355X box (run from the repo root through gpurun); summaries are copied to profiles/ afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03
rm -rf $O && mkdir -p $O
X="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 1"
S="--steps 100 --warmup 10 $X"
P="--steps 10 --warmup 2 $X"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sample -- python bench.py $S > $O/sample.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sample_b64 -- python bench.py --batch 64 $S > $O/sample_b64.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train128 -- python bench.py --mode train --batch 128 --steps 25 --warmup 5 --no-cpu-baseline --no-roofline > $O/train128.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train1024 -- python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/train1024.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py $P > $O/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py $P > $O/pmc_w.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python bench.py $P > $O/pmc_m.log 2>&1
mkdir -p $O/out
for n in sample sample_b64 train128 train1024; do cp $(find $O/$n -name "*kernel_stats.csv" | head -1) $O/out/r03_${n}_kernel_stats.csv; done
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/out/r03_gemm_traffic.json > $O/traffic.log
python tools/pmc_mfma.py $O/pmc_m $O/out/r03_mfma_util.json > $O/mfma.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python tools/prof_summary.py $O/sample 110 12; python tools/prof_summary.py $O/train128 30 12; tail -5 $O/traffic.log; head -6 $O/mfma.log
