set -o pipefail
O=gpurun_out/r03n
mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --workload candidate --steps 2 --warmup 1 --batch 8 --images 24 --no-cpu-baseline > $O/cand2.json.log 2> $O/cand2.err; echo "rc $?"
python -c "
import json
d=json.loads([l for l in open('$O/cand2.json.log') if l.startswith('{')][0])
print(d['value'], d['unit'], d['n_gpus'], d['config']['parallelism'][:40], [ (r['rank'], r['elapsed_s'], r['cpu_affinity']) for r in d['ranks']], d['collective_check'], d['fid_values'])
"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --dist-backend gloo --workload adm128 --steps 1 --warmup 1 --batch 2 --no-cpu-baseline --no-kernel-events > $O/adm128_2.json.log 2> $O/adm128_2.err; echo "rc $?"
python -c "
import json
d=json.loads([l for l in open('$O/adm128_2.json.log') if l.startswith('{')][0])
print(d['value'], d['unit'], d['n_gpus'], d['config']['global_batch'], d['output_check']['finite'])
"
tail -3 $O/cand2.err
