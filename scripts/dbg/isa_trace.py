"""Condensed memory/MFMA event trace of one kernel of a hipcc -S listing (diagnostic).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/k.s monosdf_amd/csrc/sdf_mlp.hip
  python scripts/dbg/isa_trace.py /tmp/k.s msdf_sdf_backward_k > /tmp/trace.txt

Consecutive instructions of one kind are folded ("mfma x128"); what to look for is a chain
"gload / wait vmcnt(0) / gstore" repeated per tile: one exposed memory round trip each.
"""
import re
import sys


def trace(path, kernel):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\d+%s\w*:' % kernel, l))
    # the function's end label, not its first s_endpgm: an early return (a skipped sampler round) may be laid out first
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    out, last, cnt = [], None, 0
    for l in lines[start:end]:
        s = l.strip()
        if s.startswith('v_mfma'): k = 'mfma'
        elif s.startswith('global_load_lds'): k = 'DMA'
        elif s.startswith('global_load'): k = 'gload'
        elif s.startswith('global_store'): k = 'gstore'
        elif s.startswith('s_waitcnt') and 'vmcnt' in s: k = 'wait ' + s.split(None, 1)[1]
        elif s.startswith('s_barrier'): k = 'barrier'
        elif s.startswith('scratch_'): k = s.split()[0]
        elif s.startswith('s_cbranch') or s.startswith('s_branch'): k = 'BR'
        else: continue
        if k == last: cnt += 1
        else:
            if last: out.append('%s x%d' % (last, cnt))
            last, cnt = k, 1
    out.append('%s x%d' % (last, cnt))
    return out


if __name__ == '__main__':
    print('\n'.join(trace(sys.argv[1], sys.argv[2])))
