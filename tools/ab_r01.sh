set -e
B="python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-train --no-roofline"
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run base GPT_ATT_VARIANT=0 GPT_GEMM_TAILSPLIT=0
run att1 GPT_ATT_VARIANT=1 GPT_GEMM_TAILSPLIT=0
run att2 GPT_ATT_VARIANT=2 GPT_GEMM_TAILSPLIT=0
run att3 GPT_ATT_VARIANT=3 GPT_GEMM_TAILSPLIT=0
run tail GPT_ATT_VARIANT=0 GPT_GEMM_TAILSPLIT=1
run base2 GPT_ATT_VARIANT=0 GPT_GEMM_TAILSPLIT=0
run both GPT_ATT_VARIANT=3 GPT_GEMM_TAILSPLIT=1
