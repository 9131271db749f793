#!/bin/bash
# Direct power / clock evidence for DESIGN.md 5.1.3: the SMU's socket power, per-XCD shader clocks and PPT (power
# limiter) residency sampled at 20 Hz (tools/power_sampler.c) while (a) bench.py runs 400 steps of the fp16 job,
# (b) the byte job (integer cores), (c) the MFMA-only micro-benchmark runs; plus GRBM_GUI_ACTIVE per launch of
# sweep_kernel<1> (cycles the GPU was busy / the launch's duration = its mean clock).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
[ -x tools/power_sampler ] || gcc -O2 -I/opt/rocm/include tools/power_sampler.c -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/power_sampler
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_clock tools/ubench_clock.hip 2>/dev/null
rocm-smi --showpower --showclocks --showmaxpower > $OUT/smi_idle.txt 2>&1
# (a) fp16 job, sustained
./tools/power_sampler 45 20 > $OUT/power_f16.csv 2>$OUT/power_f16.err &
SP=$!
sleep 3
timeout 300 python bench.py --steps 400 --warmup 3 --no-cpu-baseline --u8-images 0 --sustained-steps 0 > $OUT/bench_400.json 2> $OUT/bench_400.err; echo "bench400 rc=$?"
sleep 3
kill $SP 2>/dev/null; wait $SP 2>/dev/null
# (b) byte job on the integer cores, sustained
./tools/power_sampler 40 20 > $OUT/power_i8.csv 2>$OUT/power_i8.err &
SP=$!
sleep 3
timeout 300 python bench.py --workload synthetic-u8 --images 192 --desc 8192 --steps 100 --warmup 2 --no-cpu-baseline --sustained-steps 0 > $OUT/bench_i8_100.json 2> $OUT/bench_i8_100.err; echo "bench i8 rc=$?"
sleep 2
kill $SP 2>/dev/null; wait $SP 2>/dev/null
# (c) MFMA-only / VALU-only loops
./tools/power_sampler 30 20 > $OUT/power_ubench.csv 2>$OUT/power_ubench.err &
SP=$!
sleep 2
{ timeout 60 /tmp/ubench_clock; timeout 60 /tmp/ubench_clock sustain mfma 10; timeout 60 /tmp/ubench_clock sustain valu 6; } > $OUT/ubench_clock.txt 2>&1
sleep 2
kill $SP 2>/dev/null; wait $SP 2>/dev/null
# GRBM_GUI_ACTIVE per launch (its own PMC pass)
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmc_grbm
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --u8-images 0 --sustained-steps 0 > $OUT/pmc_grbm.log 2>&1; echo "grbm rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pmc_grbm_trace -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --u8-images 0 --sustained-steps 0 > $OUT/pmc_grbm_trace.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/pmc_grbm.json "sweep_kernel<1>,sweep_kernel<3>" $OUT/pmc_grbm | tail -20
python tools/power_summary.py $OUT > $OUT/power_trace.txt 2>&1; cat $OUT/power_trace.txt
find $OUT/pmc_grbm $OUT/pmc_grbm_trace -type f -size +8M -delete
