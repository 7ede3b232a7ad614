set -e
b() { timeout -k 10 300 python3 bench.py --also "" --also_modes "" "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_step'], d.get('peak_mem_gb'))" "$@"; }
b --batch 1024
b --model inception_v3_head --batch 256 --hip_graph 1 --steps 30
b --model inception_v3_head --batch 256 --hip_graph 1 --steps 30 --trunk_bn moving
b --model inception_v3_head --batch 1024 --steps 10
b --model inception_v3_head --batch 32 --hip_graph 1 --steps 60 --trunk_bn moving
b --model flower_cnn
timeout -k 10 200 python3 tools/bench_ps.py --backend xgmi 2>&1 | tail -2
timeout -k 10 200 python3 tools/bench_ps.py --backend tcp 2>&1 | tail -2
