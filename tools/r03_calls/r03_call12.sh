set -o pipefail
O=gpurun_out/r03k
mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/$tag.json.log 2>$O/$tag.err; python -c "import json; d=json.loads([l for l in open('$O/$tag.json.log') if l.startswith('{')][0]); print('$tag', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run base$i A=1
run seq$i ADM_OVERLAP_GUIDANCE=0
run sidehi$i ADM_SIDE_PRIORITY=-1
run mainhi$i ADM_MAIN_PRIORITY=-1
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --graph > $O/graph.json.log 2>$O/graph.err; python -c "import json; d=json.loads([l for l in open('$O/graph.json.log') if l.startswith('{')][0]); print('graph', d['value'], d['ms_per_step'])"
