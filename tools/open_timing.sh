#!/bin/bash
# uvaia_gpu_open_tuned with the query-side tables built by host threads (query_tables=1) and on the device (2, the default), at 10 000
# and 1 000 queries: bench.py's engine_open_s (the first open of a process: HIP's own start-up, 0.04-0.13 s, is part of it), twice.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for q in 10000 1000; do for how in 1 2; do
  v=$(timeout -k 10 300 python bench.py --queries $q --refs 65536 --pool 65536 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity $([ $q = 10000 ] && echo --mode acgt) --tuning query_tables=$how 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['config']['engine_open_s'], b['config']['query_prepare_s'], b['ms_per_step'])") || exit 1
  echo "queries $q query_tables $how: engine_open_s query_prepare_s ms_per_step = $v"
done; done; done
