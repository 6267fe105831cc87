#!/bin/bash
# time bench.py kernels for every library under build_variants/ (diagnostic A/B runs)
for f in build_variants/lib_*.so; do
  v=$(basename $f .so)
  MONOSDF_HIP_LIB=$PWD/$f python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels_ms_per_step']
print('$v', round(d['ms_per_step'],3), 'loss', round(d['loss'],6), {a:k[a] for a in ('msdf_sdf_backward','msdf_sdf_fwd_grad','msdf_sdf_forward_if','msdf_wgrad')})"
done
