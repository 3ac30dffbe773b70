for l in tools/_bin/liboffk_k16nodma.so tools/_bin/liboffk_k16now.so optical-flow-guided-feature-pytorch_amd/liboffk.so; do OFFK_LIB=$l python bench.py --no-secondary --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', round(d['stage_ms']['pw_reduce'], 4))"; done
