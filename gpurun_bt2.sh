python -m pytest tests/test_gpu_builder.py -x -q 2>&1 | tail -1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('gpu-core', d['value'], d['config']['blas_build_ms'], d['config']['bvh_build_s'])"; done
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --cpu-build 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('cpu', d['value'], d['config']['blas_build_ms'], d['config']['bvh_build_s'])"; done
