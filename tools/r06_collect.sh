# copy what tools/r06_measure.sh left under gpurun_out/r06m into profiles/ (tracked): bench lines, kernel statistics, PMC passes,
# traffic entries, e2e files, timelines, the test log.  Run in the repository root after the gpurun call has merged its output.
O=gpurun_out/r06m
for t in default repeats k51 octoploid hexaploid; do
  grep '^{' $O/bench_$t.log > profiles/r06_bench_$t.json.log
  cp "$(find $O/stats_$t -name '*kernel_stats.csv' | head -1)" profiles/r06_kernel_stats_$t.csv
  { for p in a b c d; do [ -f $O/pmc_${t}_$p.txt ] && { echo "# pass $p"; cat $O/pmc_${t}_$p.txt; }; done; } > profiles/r06_pmc_$t.txt
done
grep '^{' $O/bench_k30.log > profiles/r06_bench_k30.json.log
grep '^{' $O/bench_k51_G1e9.log > profiles/r06_bench_k51_G1e9.json.log
cp $O/hbm_traffic.json profiles/hbm_traffic.json
cp $O/e2e_config3_vshards8.json profiles/r06_e2e_config3_virtual_shards_8_and_out_of_core.json
cp $O/step_timeline_uniform.txt profiles/r06_step_timeline_uniform_G1e9.txt
cp $O/step_timeline_hexaploid.txt profiles/r06_step_timeline_hexaploid.txt
cp $O/pytest_full.txt profiles/r06_pytest_gpu_full.txt
grep '^{' $O/bench_forced.log > profiles/r06_bench_forced_exchange_G1.25e8_bm29.json.log
grep '^{' $O/bench_forced_replay.log > profiles/r06_bench_forced_exchange_G1.25e8_bm29_replay.json.log
python - <<'PY'
import json, csv, glob
for t in ('default', 'repeats', 'k51', 'octoploid', 'hexaploid'):
    d = json.load(open(f'profiles/r06_bench_{t}.json.log')); r = d['roofline']
    print(t, 'ms/step %.3f' % d['ms_per_step'], 'value %.4g' % d['value'], 'frac %.3f' % r['frac'], 'p1 %.3f alone %.3f' % (r['kernel_ms']['ms_pass1'], r['pass1_kernel_alone_ms']),
          'lookup %.3f (part %.3f)' % (r['kernel_ms']['ms_rclookup'], r['requests']['ms_partition']), 'p2 %.3f' % r['kernel_ms']['ms_pass2'],
          'traffic', r['traffic'] and round(r['traffic'] / 1e9, 2), 'parity', d['parity']['ok'], 'whole-job %.3f' % r['whole_job_frac_of_the_B_alg_roofline'])
    f = glob.glob(f'gpurun_out/r06m/stats_{t}/*kernel_stats.csv')[0]
    for row in csv.DictReader(open(f)):
        n = row['Name']
        if any(k in n for k in ('kf_pass1_d', 'kl_part', 'kl_probe', 'kf_pass2<', 'kf_bigfix')):
            print('      %-50s %8.3f ms' % (n.replace('void ', '')[:50], float(row['AverageNs']) / 1e6))
PY
