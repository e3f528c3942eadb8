export AFX_BENCH_UNCHECKED=1
LIB=aeonflux_amd/lib/libaeonflux_gpu.so
cp $LIB /tmp/shipped.so
for which in /tmp/shipped.so variants/alias_tables.so /tmp/shipped.so variants/alias_tables.so; do
  cp $which $LIB
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); v=d['valu']; r=d['roofline']
        print('$which', round(d['ms_per_step'],1), 'window', r['kernels_ms_per_step']['k_msm_window'], 'clock MHz', round(v['core_clock_mhz_measured']))"
done
cp /tmp/shipped.so $LIB
(python bench.py --steps 30 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &) ; sleep 12; for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Power" | tr "\n" " "; echo; sleep 2; done; sleep 6
(timeout 60 variants/mad_sustained > /dev/null &) ; sleep 0.3; 
for i in 1 2 3; do variants/mad_sustained > /dev/null & sleep 0.25; rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Power" | tr "\n" " "; echo; wait; done
