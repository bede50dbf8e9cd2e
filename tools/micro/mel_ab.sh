# Same-run A/B of whole-library builds through bench.py (C2 step, core launch time from HIP events, the rest = front end
# + launch gaps):  LIBS="tools/micro/bin/libkm_mel0.so ..." bash tools/micro/mel_ab.sh
for l in ${LIBS}; do KM_LIBRARY=$l python bench.py --steps ${STEPS:-50} --cpu-seconds 0 --no-split ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s step %.1f us' % ('$l'.split('/')[-1], d['ms_per_step']*1e3) + (' core %.1f us rest %.1f us' % (r['launch_ms']*1e3, d['ms_per_step']*1e3-r['launch_ms']*1e3) if 'launch_ms' in r else ''))"; done
