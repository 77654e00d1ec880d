set -o pipefail
O=gpurun_out/r03t
mkdir -p $O
python -m pytest tests/test_bench_multi.py tests/test_bench_host.py -q -x -k "two_ranks or host or pin or slice or guided" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --dist-backend gloo --steps 1 --warmup 1 --batch 4 --no-cpu-baseline --no-kernel-events > $O/two.json.log 2>$O/two.err
python -c "
import json
d=json.loads([l for l in open('$O/two.json.log') if l.startswith('{')][0])
print([(r['rank'], r['pci_bus_id'], r['cpu_affinity']) for r in d['ranks']])"
cat /sys/bus/pci/devices/0000:0d:00.0/local_cpulist 2>/dev/null || true
