# round-6 measurement set on ONE box: the full -m gpu suite, then per workload the kernel stats and the PMC passes (never combined
# with tracing), the traffic entry with the library's hash, and ONLY THEN the bench line -- so that every kept line carries its own
# `traffic` -- then e2e, timeline, k = 51 at 1 Gbp.  Usage on the GPU box: bash tools/r06_measure.sh [part ...]   (default: all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R
export TMPDIR=/tmp
PARTS=${@:-"suite default repeats k51 oct hex k30 timeline k51big vshards forced"}
J=$R/profiles/hbm_traffic.json
line() { grep '^{' $1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], {k: round(v,3) for k,v in r["kernel_ms"].items()}, "p1alone %.3f" % r["pass1_kernel_alone_ms"], "frac %.3f" % r["frac"], "traffic", r["traffic"], "parity", d["parity"]["ok"], "cpu", (d.get("cpu_baseline") or {}).get("value"))' $2; }
prof() {   # prof <tag> <traffic-key> <bench args...>
  tag=$1; key=$2; shift 2
  B="python $R/bench.py --steps 3 --warmup 1 --no-cpu $*"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o r06 -- $B > $O/stats_$tag.log 2>&1
    timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${tag}_c -o r06 -- $B > $O/pmc_${tag}_c.log 2>&1
    timeout 400 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_${tag}_d -o r06 -- $B > $O/pmc_${tag}_d.log 2>&1 )
  for d in pmc_${tag}_c pmc_${tag}_d; do python tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; find $O/$d -name '*counter_collection.csv' -delete; done
  find $O/stats_$tag -name '*kernel_trace.csv' -delete
  n=$(grep '^{' $O/stats_$tag.log | python -c 'import sys,json,re; d=json.loads(sys.stdin.read()); print(re.search(r": (\d+) table entries", d["config"]["workload"]).group(1))')
  python tools/make_traffic.py $key $n $O/pmc_${tag}_c.txt $O/pmc_${tag}_d.txt "profiles/r06_pmc_${tag}.txt" $J
  ( timeout 600 python bench.py "$@" ) > $O/bench_$tag.log 2>&1; line $O/bench_$tag.log $tag
}
for part in $PARTS; do case $part in
suite)
  ( timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu ) > $O/pytest_full.txt 2>&1
  grep "passed\|failed\|error" $O/pytest_full.txt | tail -3
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/pytest_full.txt ;;
default)
  prof default k31
  B="python $R/bench.py --steps 3 --warmup 1 --no-cpu"
  ( cd /tmp
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_default_a -o r06 -- $B > $O/pmc_default_a.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_default_b -o r06 -- $B > $O/pmc_default_b.log 2>&1 )
  for d in pmc_default_a pmc_default_b; do python tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; find $O/$d -name '*counter_collection.csv' -delete; done ;;
repeats) prof repeats k31_repeats --workload repeats ;;
k51)     prof k51 k51 --k 51 --genome 5e8 ;;
oct)     prof octoploid k31_octoploid --workload octoploid ;;
hex)     prof hexaploid k51_hexaploid --workload hexaploid ;;
k30)     ( timeout 300 python bench.py --k 30 --no-cpu ) > $O/bench_k30.log 2>&1; line $O/bench_k30.log k30 ;;
k51big)  ( timeout 900 python bench.py --k 51 --genome 1e9 --no-cpu --steps 5 --warmup 1 ) > $O/bench_k51_G1e9.log 2>&1; line $O/bench_k51_G1e9.log k51_G1e9 ;;
e2e)
  # (the default bench line carries the end-to-end block since round 6: bench.py runs both programs itself; this part keeps the
  #  breakdown of the executable's wall time in both process modes, reference skipped)
  ( E2E_SKIP_REF=1 timeout 900 python tools/e2e_config12.py 4e8 diploid skipT1 ) > $O/e2e_1e9.json 2> $O/e2e_err.txt
  grep -A3 "end_to_end_T" $O/e2e_1e9.json | grep -v engine_line | cut -c1-200 ;;
timeline)
  for w in uniform hexaploid; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$w -o r06 -- python $R/bench.py --workload $w --steps 4 --warmup 2 --no-cpu > $O/tl_$w.log 2>&1 )
    python tools/step_timeline.py $O/tl_$w kf_pass1_d > $O/step_timeline_$w.txt 2>&1; rm -rf $O/tl_$w
    tail -n 2 $O/step_timeline_$w.txt
  done ;;
forced)   # one rank's share of an 8-GPU step: an eighth of the table, every collective forced in a one-rank group, 29-bit map
  ( SMG_FORCE_EXCHANGE=1 SMG_BM_BITS=29 timeout 300 python bench.py --genome 1.25e8 --no-cpu --steps 20 --warmup 3 ) > $O/bench_forced.log 2>&1; line $O/bench_forced.log forced_G1.25e8_bm29
  ( SMG_FORCE_EXCHANGE=1 SMG_BM_BITS=29 SMG_REPLAY=1 timeout 300 python bench.py --genome 1.25e8 --no-cpu --steps 20 --warmup 3 ) > $O/bench_forced_replay.log 2>&1; line $O/bench_forced_replay.log forced_replay ;;
vshards)
  ( E2E_SKIP_REF=1 E2E_VSHARDS=8 timeout 900 python tools/e2e_full_table.py uniform ) > $O/e2e_config3_vshards8.json 2> $O/e2e_vs_err.txt
  grep -v engine_line $O/e2e_config3_vshards8.json | head -40; grep "smg\]" $O/e2e_config3_vshards8.json | cut -c1-400 ;;
esac; done
cp $J $O/hbm_traffic.json
