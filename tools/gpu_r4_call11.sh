#!/bin/bash
# Round 4, call 11: SQ / TCC counters of the tail kernels (unpipelined bench command): what are prune, slot assignment and the exact re-check waiting for?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
cd /tmp; rm -rf $OUT/pmc_tail_sq $OUT/pmc_tail_tcc
MSFM_PIPELINE=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tail_sq -- $BENCH > $OUT/pmc_tail_sq.log 2>&1; echo "sq rc=$?"
MSFM_PIPELINE=1 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tail_tcc -- $BENCH > $OUT/pmc_tail_tcc.log 2>&1; echo "tcc rc=$?"; tail -3 $OUT/pmc_tail_tcc.log
cd $ROOT
KERN="sweep_i8_kernel<1>,sweep_kernel<3>,pf_prune_q8_kernel,pf_assign_kernel,pf_exact_candidates_kernel,epilogue_kernel,fill_segs_kernel"
python tools/pmc_summary.py $OUT/r4_pmc_tail.json "$KERN" $OUT/pmc_tail_sq $OUT/pmc_tail_tcc > /dev/null 2>&1; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_pmc_tail.json"))
for k,v in d.items():
    if k.startswith("_"): continue
    g={c:x.get("per_launch_mean",x.get("per_launch_KB_mean")) for c,x in v.items()}
    print(k, {c:("%.3g"%x) for c,x in g.items()})
PY
find $OUT/pmc_tail_sq $OUT/pmc_tail_tcc -type f -size +8M -delete
