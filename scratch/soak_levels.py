# Soak: the fuzz sweep of tests/gpu_util.py under the contexts of tests/test_gpu_levels.py, more seeds.
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tests'))
from gpu_util import fresh_context, fuzz_configurations
envs = [dict(MI_TRACER_TAIL=0), dict(MI_TRACER_TAIL=2000), dict(MI_TRACER_TAIL=2000, MI_TRACER_SPLIT_MIN=1000), dict(MI_TRACER_STREAMS=1), dict(MI_TRACER_TAIL=300, MI_TRACER_SPLIT_MIN=500)]
n = 0
for env in envs:
    with fresh_context(**env) as c:
        for seed in range(300, 312):
            fuzz_configurations(c, seed, 40)
            n += 40
        print(env, "ok", flush=True)
print("configurations", n)
