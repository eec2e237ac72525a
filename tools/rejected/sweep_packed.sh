# packed (embed_fwd_packed) vs one-feature-per-instruction (embed_fwd_uniform) mapping, bench.py timing
for cfg in "c2 uniform" "c2 zipf" "c5 uniform" "c3 uniform"; do set -- $cfg
for np in 0 1; do if [ $np = 1 ]; then export NRX_NO_PACKED=1; else unset NRX_NO_PACKED; fi
echo -n "$1 ids=$2 $( [ $np = 1 ] && echo uniform-kernel || echo packed-kernel ): "; python bench.py --workload $1 --ids $2 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%.1f us  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"
done; done
