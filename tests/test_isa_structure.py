"""Structure of the generated gfx950 code of the fused MLP kernels (CPU-only: hipcc cross-compiles, nothing runs).

What is guarded: an epilogue written as "for every tile: load, compute, store" compiles to one
global_load -> s_waitcnt vmcnt(0) -> global_store round trip per tile, because the store may alias the next load as
far as the compiler knows.  Round 2 found that in every epilogue of the SDF / colour backward kernels (16 exposed
memory latencies per matrix product) and removed it; this test fails if such a chain comes back."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts', 'dbg'))
import isa_trace  # noqa: E402

HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'

KERNELS = {
    'sdf_mlp.hip': ('msdf_sdf_forward_k', 'msdf_sdf_fwd_grad_k', 'msdf_sdf_backward_k'),
    'color_mlp.hip': ('msdf_color_forward_k', 'msdf_color_backward_k'),
}


def _longest_chain(events):
    ev = [e for e in events if not e.startswith('BR')]
    best = cur = i = 0
    while i + 2 < len(ev):
        if ev[i].startswith('gload') and ev[i + 1].startswith('wait vmcnt(0)') and ev[i + 2].startswith('gstore'):
            cur += 1
            i += 3
        else:
            best, cur = max(best, cur), 0
            i += 1
    return max(best, cur)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
@pytest.mark.parametrize('source', sorted(KERNELS))
def test_no_load_wait_store_chain_per_tile(source, tmp_path):
    listing = str(tmp_path / (source + '.s'))
    subprocess.run([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-value', '-S',
                    '--cuda-device-only', '-o', listing, os.path.join(ROOT, 'monosdf_amd', 'csrc', source)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    for kernel in KERNELS[source]:
        events = isa_trace.trace(listing, kernel)
        assert any(e.startswith('mfma') for e in events), kernel          # the listing really is the kernel
        assert _longest_chain(events) <= 2, (kernel, _longest_chain(events))
