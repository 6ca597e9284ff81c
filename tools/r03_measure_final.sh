# round-3 measurement set: parity, bench lines, kernel stats, PMC passes (never combined with tracing), forced exchange, e2e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
( timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu ) > $O/pytest_full.txt 2>&1
grep -n "passed\|failed" $O/pytest_full.txt | tail -2
( timeout 400 python bench.py ) > $O/bench_default.log 2>&1
grep "^{" $O/bench_default.log | cut -c1-500
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03 -- $B > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_a -o r03 -- $B > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_b -o r03 -- $B > $O/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c -o r03 -- $B > $O/pmc_c.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_d -o r03 -- $B > $O/pmc_d.log 2>&1
B51="python $R/bench.py --k 51 --genome 5e8 --steps 3 --warmup 1 --no-cpu"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats51 -o r03 -- $B51 > $O/stats51.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc51_c -o r03 -- $B51 > $O/pmc51_c.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc51_d -o r03 -- $B51 > $O/pmc51_d.log 2>&1
BR="python $R/bench.py --workload repeats --steps 3 --warmup 1 --no-cpu"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsrep -o r03 -- $BR > $O/statsrep.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmcrep_c -o r03 -- $BR > $O/pmcrep_c.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmcrep_d -o r03 -- $BR > $O/pmcrep_d.log 2>&1
cd $R
for d in pmc_a pmc_b pmc_c pmc_d pmc51_c pmc51_d pmcrep_c pmcrep_d; do python tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; find $O/$d -name '*counter_collection.csv' -delete; done
find $O -name '*kernel_trace.csv' -delete
( timeout 300 python bench.py --k 51 --genome 5e8 ) > $O/bench_k51.log 2>&1
grep "^{" $O/bench_k51.log | cut -c1-300
( timeout 300 python bench.py --workload repeats --no-cpu ) > $O/bench_repeats.log 2>&1
grep "^{" $O/bench_repeats.log | cut -c1-300
( timeout 300 python bench.py --k 30 --no-cpu ) > $O/bench_k30.log 2>&1
grep "^{" $O/bench_k30.log | cut -c1-200
for g in 1e9 5e8 2.5e8 1.25e8; do
  ( SMG_FORCE_EXCHANGE=1 timeout 300 python bench.py --genome $g --no-cpu --steps 10 --warmup 2 ) > $O/forced_$g.log 2>&1
  ( timeout 300 python bench.py --genome $g --no-cpu --steps 10 --warmup 2 ) > $O/plain_$g.log 2>&1
  echo "G=$g forced $(grep '^{' $O/forced_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["roofline"]["kernel_ms"])') plain $(grep '^{' $O/plain_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))')"
done > $O/rank_share.txt 2>&1
( SMG_BM_BITS=29 SMG_FORCE_EXCHANGE=1 timeout 300 python bench.py --genome 1.25e8 --no-cpu --steps 10 --warmup 2 ) > $O/forced_1.25e8_bm29.log 2>&1
echo "G=1.25e8 forced, 29-bit map (what 8 ranks use) $(grep '^{' $O/forced_1.25e8_bm29.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["roofline"]["kernel_ms"])')" >> $O/rank_share.txt
cat $O/rank_share.txt
cd /tmp
SMG_BM_BITS=29 SMG_FORCE_EXCHANGE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tf -o t -- python $R/bench.py --genome 1.25e8 --no-cpu --steps 4 --warmup 2 > $O/tf.log 2>&1
python $R/tools/step_timeline.py $O/tf kf_pass1_d > $O/timeline_forced_1.25e8_bm29.txt 2>&1
find $O -name '*kernel_trace.csv' -delete
tail -1 $O/timeline_forced_1.25e8_bm29.txt
cd $R
( timeout 900 python tools/e2e_config12.py 4e8 diploid skipT1 ) > $O/e2e_1e9.json 2> $O/e2e_err.txt
grep -A6 "end_to_end_T4" $O/e2e_1e9.json | cut -c1-600
( timeout 300 python tools/decode_time.py 4e8 31 ) > $O/decode_time.txt 2>&1; tail -1 $O/decode_time.txt
