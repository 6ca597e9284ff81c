#!/bin/bash
# The A/B blocks c2 ... c15 of profiles/r06_pass1_experiments.txt and profiles/r06_lookup_experiments.txt, as they were run
# (one gpurun call = one box per block).  Variants are built HERE with tools/build_r06.sh (one -D each; build/r06 travels with the
# tree), then compared on ONE device-resident bench table by tools/ab_libs.py (fresh engine per spec, whole runs, every plot
# checked against the first spec's and against the reference binary's golden).  Several of the switches named below existed only
# on the experiment's own source state (LDS swizzle D_SWZ, shared buckets SMG_PB_SHARE / SMG_PB_COARSE, PB_PREFETCH, count-aware
# detection, owner-major offsets): the records say which commit-state they belong to; what is left in the tree is L_ABL_NOSETP,
# L_ABL_EXTRAROW, L_ABL_NOLOOKUP, L_NB_FORCE (smg_lookup.hpp / smg_fast.hpp), D_TAILB, D_ABL (smg_pass1d.hpp) and -DSMG_TUNING.
#
#   tools/build_r06.sh base="" nosetp="-DL_ABL_NOSETP" extrarow="-DL_ABL_EXTRAROW" nb9="-DL_NB_FORCE=9" nolook="-DL_ABL_NOLOOKUP" \
#                      tailb4="-DD_TAILB=4" tailb2="-DD_TAILB=2" tailb12="-DD_TAILB=12" tune="-DSMG_TUNING"
B=build/r06
set -x
# c2: where the polyploid probe kernel's time goes; what one more DRAM row per look-up costs (the 8-byte request record)
python tools/ab_libs.py hexaploid $B/libsmg_base.so $B/libsmg_nosetp.so $B/libsmg_extrarow.so $B/libsmg_nb9.so $B/libsmg_nolook.so $B/libsmg_base.so
python tools/ab_libs.py octoploid $B/libsmg_base.so $B/libsmg_nosetp.so $B/libsmg_extrarow.so $B/libsmg_nolook.so $B/libsmg_base.so
# c3: is kl_part's time per run or per engine?  (ten runs per engine, every run printed)
AB_PER_RUN=1 AB_RUNS=10 python tools/ab_libs.py uniform $B/libsmg_base.so $B/libsmg_base.so $B/libsmg_base.so
# c2 / c14: the lists' addresses (tuning build, SMG_DEBUG=1), and the translation counters per dispatch
SMG_DEBUG=1 python tools/ab_libs.py hexaploid $B/libsmg_tune.so $B/libsmg_tune.so $B/libsmg_tune.so
( cd /tmp && rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/tlb -o tlb -- \
    python $OLDPWD/tools/ab_libs.py uniform $B/libsmg_base.so $B/libsmg_base.so $B/libsmg_base.so )
# c4 - c6: the detection pass (batch size; the final state is D_TAILB = 6 without count checks)
python tools/ab_libs.py uniform $B/libsmg_base.so $B/libsmg_tailb4.so $B/libsmg_tailb2.so $B/libsmg_tailb12.so $B/libsmg_base.so
python tools/ab_libs.py repeats $B/libsmg_base.so $B/libsmg_tailb4.so $B/libsmg_base.so
# randomised soak of the final library (three seeds in parallel)
for s in 61 62 63; do python tools/soak.py 500 $s & done; wait
