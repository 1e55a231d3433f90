#!/bin/bash
# Round-6 final evidence, ONE lease: full GPU suite, smoke, rocprofv3 profile of the DiT step (trace + PMC), one-lane VAE trace + PMC + per-shape view,
# two-lane VAE timing, and the full default bench line — all from the same build on the same box.
export TMPDIR=/tmp
OUT=gpurun_out/r06final
mkdir -p $OUT
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | grep "fullsize\]\|passed\|failed\|Warning\|Error\|attention\]" | cut -c1-1800 > $OUT/gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1
bash tools/profile_dit.sh r06final/dit > $OUT/profile_dit.log 2>&1
cp gpurun_out/r06final/dit_summary.json $OUT/dit_step.json 2>/dev/null; cp gpurun_out/r06final/dit_summary.md $OUT/dit_step.md 2>/dev/null
bash tools/profile_vae.sh r06final/vae1 > $OUT/profile_vae.log 2>&1
python tools/summarize_trace_by_grid.py $OUT/vae1/trace $OUT/vae1_by_grid.md > /dev/null 2>&1
python tools/gpu_vae_bench.py --reps 5 --lanes 2 --out $OUT/vae_two_lanes.json > $OUT/vae_two_lanes.log 2>&1
find $OUT -name "*.csv" -size +8M -delete
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/gputests.log; cat $OUT/smoke.log | tail -1; grep -h seconds $OUT/vae_two_lanes.log | cut -c1-120
python - <<'P'
import json
l=json.loads([x for x in open('gpurun_out/r06final/bench.json') if x.startswith('{')][-1])
print(json.dumps({k:l[k] for k in ("value","ms_per_step","roofline","kernel_tflops","value_data_independent","mfma_frac_whole_step")})[:1500])
print(json.dumps(l.get("vae"))[:800]); print(json.dumps(l.get("clip"))[:1200]); print(json.dumps(l.get("windows"))[:500]); print(json.dumps(l.get("cfg_step")))
P
