export STARKPERP_BENCH_SHARE_GPU=1 STARKPERP_WINDOW_BITS=16
for L in 19 21; do
  S=$(date +%s.%N)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2951$((L-12)) bench.py --gpus 8 --workload airfri --window-bits 0 --no-extras --no-cpu-baseline --steps 1 --warmup 0 --log-rows $L > gpurun_out/shard8_$L.json 2> gpurun_out/shard8_$L.err
  E=$(date +%s.%N)
  echo "log-rows $L per rank: rc=$? wall $(python -c "print(round($E-$S,1))") s"
  python -c "
import json
ls=[l for l in open('gpurun_out/shard8_$L.json') if l.startswith('{')]
if not ls: print(open('gpurun_out/shard8_$L.err').read()[-1500:])
else:
    d=json.loads(ls[0]); print(d['value'],d['ms_per_step'],d['config'].get('rows_total'),d.get('sharded_roots_match_single_gpu'),json.dumps(d['config'].get('exchange'))[:700])"
done
