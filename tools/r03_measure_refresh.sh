# refresh of the evidence that the last changes touch: bench lines, kernel stats, rank shares
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
( timeout 400 python bench.py ) > $O/bench_default.log 2>&1
grep "^{" $O/bench_default.log | cut -c1-200
( timeout 300 python bench.py --workload repeats --no-cpu ) > $O/bench_repeats.log 2>&1
grep "^{" $O/bench_repeats.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsrep -o r03 -- python $R/bench.py --workload repeats --steps 3 --warmup 1 --no-cpu > $O/statsrep.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmcrep_c -o r03 -- python $R/bench.py --workload repeats --steps 3 --warmup 1 --no-cpu > $O/pmcrep_c.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmcrep_d -o r03 -- python $R/bench.py --workload repeats --steps 3 --warmup 1 --no-cpu > $O/pmcrep_d.log 2>&1
cd $R
for d in pmcrep_c pmcrep_d; do python tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; find $O/$d -name '*counter_collection.csv' -delete; done
find $O -name '*kernel_trace.csv' -delete
for g in 1e9 5e8 2.5e8 1.25e8; do
  ( SMG_FORCE_EXCHANGE=1 timeout 300 python bench.py --genome $g --no-cpu --steps 10 --warmup 2 ) > $O/forced_$g.log 2>&1
  ( timeout 300 python bench.py --genome $g --no-cpu --steps 10 --warmup 2 ) > $O/plain_$g.log 2>&1
  echo "G=$g forced $(grep '^{' $O/forced_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["roofline"]["kernel_ms"])') plain $(grep '^{' $O/plain_$g.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))')"
done > $O/rank_share.txt 2>&1
( SMG_BM_BITS=29 SMG_FORCE_EXCHANGE=1 timeout 300 python bench.py --genome 1.25e8 --no-cpu --steps 10 --warmup 2 ) > $O/forced_1.25e8_bm29.log 2>&1
echo "G=1.25e8 forced, 29-bit map (what 8 ranks use) $(grep '^{' $O/forced_1.25e8_bm29.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["roofline"]["kernel_ms"])')" >> $O/rank_share.txt
cat $O/rank_share.txt
