import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.1f  %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
for k in ('sustained', 'sharp_state', 'hash_grid', 'alt_matrix_core', 'alt_matrix_core_x6'):
    if k in d:
        print(' ', k, '%.1f  %.3f ms' % (d[k]['value'], d[k]['ms_per_step']))
