run() { echo "== $*"; env "$@" python bench.py --no-roofline --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do run $v; done
