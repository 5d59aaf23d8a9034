# A/B of one build under two settings of an environment variable: bash tools/scripts/ab_env.sh NAME VALUE_A VALUE_B
cd $GRAFT_REPO_ROOT
one() { env $1=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$2', d['value'], d['ms_per_step'], d['stage_ms'], d.get('verified_vs_oracle'))"; }
for i in 1 2; do one $1 $2; one $1 $3; done
