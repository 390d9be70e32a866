for b in 200 400 800; do for t in 6 8; do
 echo "== batch $b threads $t"; LSN_DECODE_THREADS=$t timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --batch $b 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); d=j['detail']['per_step']; print(j['value'], 'ms/step', j['ms_per_step'], {k:round(v,1) for k,v in d.items() if k.startswith('ms_')}, 't128', j['detail']['kernel_ms_per_step']['k_turbo<128>'], 't64', j['detail']['kernel_ms_per_step']['k_turbo<64>'])"
done; done
