"""ctypes binding of libmonosdf_hip.so (C ABI declared in include/monosdf_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is
missing or a kernel launch fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MONOSDF_HIP_LIB: another build of the same library (kernel tuning experiments), never a different backend
LIB_PATH = os.environ.get('MONOSDF_HIP_LIB') or os.path.join(_HERE, 'libmonosdf_hip.so')

MAX_LAYERS = 10
MAX_TILES = 17


class Layer(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('kt', 'ot', 'wf_off', 'wb_off', 'bias_off', 'skip_tile', 'hpre', 'qpre', 'abpre',
                 'ktp', 'otp', 'pad_')]


class Plan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('n_layers', 'e_tiles', 'aux_tiles', 'n_freqs', 'sdf_slot', 'feat_tiles', 'hsum',
                 'qsum', 'absum', 'wsdf_off', 'mode', 'out_act', 'precision', 'out_rows')] + [('layer', Layer * MAX_LAYERS)]


class PackRule(C.Structure):
    _fields_ = [('w_off', C.c_int32), ('b_off', C.c_int32), ('rows', C.c_int32), ('cols', C.c_int32),
                ('rowmap_off', C.c_int32), ('colmap_off', C.c_int32), ('scale', C.c_float),
                ('pad_', C.c_int32)]


class WgradItem(C.Structure):
    _fields_ = [('x', C.c_int64), ('y', C.c_int64), ('v', C.c_int64),
                ('part_off', C.c_int64), ('colsum_off', C.c_int64), ('vrow_off', C.c_int64),
                ('x_ld', C.c_int32), ('y_ld', C.c_int32), ('wx', C.c_int32), ('wy', C.c_int32),
                ('n_splits', C.c_int32), ('bufs', C.c_int32)]


class ReduceRule(C.Structure):
    _fields_ = [('part_off', C.c_int64), ('dst_off', C.c_int64), ('n_blocks', C.c_int32),
                ('wx', C.c_int32), ('wy', C.c_int32), ('rowmap_off', C.c_int32),
                ('colmap_off', C.c_int32), ('dst_ld', C.c_int32), ('fixed_row', C.c_int32),
                ('scale', C.c_float)]


_P = C.c_void_p


class FgArgs(C.Structure):
    _fields_ = [('wpack', _P), ('bpack', _P), ('x', _P), ('aux', _P),
                ('P', C.c_int32), ('P_pad', C.c_int32), ('n_clamp', C.c_int32), ('n_feat', C.c_int32),
                ('clamp_radius', C.c_float), ('sphere_scale', C.c_float),
                ('sdf', _P), ('feat', _P), ('nrm', _P), ('r_aux', _P), ('clamped', _P),
                ('H', _P), ('PM', _P), ('IN0', _P), ('save', C.c_int32), ('aux_C', C.c_int32), ('aux_LC', C.c_int32),
                ('aux_dx_scale', C.c_float), ('dy_dx', _P)]


class BwArgs(C.Structure):
    _fields_ = [('wpack', _P), ('bpack', _P), ('x', _P),
                ('P', C.c_int32), ('P_pad', C.c_int32), ('n_feat', C.c_int32), ('n_split', C.c_int32),
                ('g_sdf', _P), ('g_feat', _P), ('g_nrm', _P), ('g_raux', _P), ('clamped', _P),
                ('H', _P), ('PM', _P), ('QB', _P), ('T', _P), ('AB', _P), ('GSDF', _P), ('QLAST', _P),
                ('g_aux', _P), ('g_sdf_b', _P), ('g_nrm_b', _P), ('aux_C', C.c_int32), ('aux_LC', C.c_int32),
                ('dy_dx', _P), ('gg_out', _P), ('aux_dx_scale', C.c_float), ('pad_', C.c_int32)]


class ColorFwdArgs(C.Structure):
    _fields_ = [('wpack', _P), ('bpack', _P), ('x', _P), ('dirs', _P), ('nrm', _P), ('feat', _P),
                ('code', _P), ('P', C.c_int32), ('P_pad', C.c_int32), ('spr', C.c_int32),
                ('save', C.c_int32), ('rgb', _P), ('H', _P), ('MISC', _P)]


class ColorBwdArgs(C.Structure):
    _fields_ = [('wpack', _P), ('bpack', _P), ('rgb', _P), ('g_rgb', _P),
                ('P', C.c_int32), ('P_pad', C.c_int32), ('H', _P), ('AB', _P), ('g_feat', _P),
                ('g_misc', _P), ('g_nrm', _P)]


class CompositeArgs(C.Structure):
    _fields_ = [('z', _P), ('sdf', _P), ('rgb', _P), ('nrm', _P), ('beta', _P), ('depth_scale', _P),
                ('N', C.c_int32), ('S', C.c_int32), ('white_bkgd', C.c_int32),
                ('bg0', C.c_float), ('bg1', C.c_float), ('bg2', C.c_float),
                ('weights', _P), ('rgb_values', _P), ('depth_values', _P), ('normal_map', _P),
                ('wsum', _P), ('pose', _P), ('pose_stride', C.c_int32), ('depth_scale_stride', C.c_int32),
                ('depth_vals', _P)]


class CompositeBwdArgs(C.Structure):
    _fields_ = [('z', _P), ('sdf', _P), ('rgb', _P), ('nrm', _P), ('beta', _P), ('depth_scale', _P),
                ('weights', _P), ('wsum', _P), ('depth_values', _P), ('g_rgb_values', _P),
                ('g_depth', _P), ('g_normal', _P), ('g_weights', _P),
                ('N', C.c_int32), ('S', C.c_int32), ('white_bkgd', C.c_int32),
                ('bg0', C.c_float), ('bg1', C.c_float), ('bg2', C.c_float),
                ('g_sdf', _P), ('g_rgb', _P), ('g_nrm', _P), ('g_beta_part', _P),
                ('pose', _P), ('pose_stride', C.c_int32), ('depth_scale_stride', C.c_int32)]


class WnLayer(C.Structure):
    _fields_ = [('v', _P), ('g', _P), ('b', _P), ('rows', C.c_int32), ('cols', C.c_int32),
                ('w_off', C.c_int32), ('b_off', C.c_int32), ('row_off', C.c_int32), ('has_g', C.c_int32)]


class ProbeLossArgs(C.Structure):
    _fields_ = [('rgb', _P), ('nrm', _P), ('depth', _P), ('g1', _P), ('g2', _P), ('N', C.c_int32), ('M', C.c_int32),
                ('w_normal', C.c_float), ('w_depth', C.c_float), ('w_eik', C.c_float), ('w_smooth', C.c_float),
                ('g_rgb', _P), ('g_nrm', _P), ('g_depth', _P), ('g_g1', _P), ('g_g2', _P), ('partial', _P)]


class MonoSdfLossArgs(C.Structure):
    _fields_ = [(n, _P) for n in ('rgb', 'depth', 'normal', 'sdf', 'grad_theta', 'grad_nei', 'rgb_gt', 'depth_gt',
                                  'normal_gt', 'mask_gt')] + \
               [(n, C.c_int32) for n in ('N', 'S', 'E', 'gamma', 'scale_invariant')] + \
               [(n, C.c_float) for n in ('w_eik', 'w_smooth', 'w_depth', 'w_nl1', 'w_ncos')] + \
               [(n, _P) for n in ('mask', 'out', 'g_rgb', 'g_depth', 'g_normal', 'g_theta', 'g_nei')]


class SamplerArgs(C.Structure):
    _fields_ = [('ray_o', _P), ('ray_d', _P), ('N', C.c_int32), ('M', C.c_int32), ('m_max', C.c_int32),
                ('n_eval', C.c_int32), ('n_final', C.c_int32), ('n_extra', C.c_int32),
                ('round_idx', C.c_int32), ('max_rounds', C.c_int32), ('training', C.c_int32),
                ('beta_iters', C.c_int32), ('near', C.c_float), ('far', C.c_float), ('bound', C.c_float),
                ('eps', C.c_float), ('add_tiny', C.c_float), ('lemma', C.c_float),
                ('beta0', _P), ('z', _P), ('sdf', _P), ('new_z', _P), ('new_sdf', _P), ('new_pos', _P),
                ('pts', _P), ('beta', _P), ('flags', _P), ('jitter', _P), ('u_final', _P), ('final_z', _P),
                ('extra_idx', _P), ('eik_idx', _P), ('z_out', _P), ('z_eik', _P), ('pts_out', _P),
                ('eik_uniform', _P), ('nei_rand', _P), ('far_out', _P), ('rounds_out', _P),
                ('dbg_dstar', _P), ('dbg_err0', _P), ('dbg_cdf', _P), ('eik_u', _P), ('eik_unit', C.c_int32),
                ('pad_', C.c_int32)]


_ERR = {1: 'invalid argument', 2: 'kernel launch failed', 3: 'unsupported configuration'}

# name -> argtypes (restype is always int)
_SIGNATURES = {
    'msdf_abi_version': [],
    'msdf_hash_encode_forward': [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_float, C.c_uint32, C.c_int, _P, _P],
    'msdf_hash_encode_backward': [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_float, C.c_uint32, C.c_int, _P, _P, _P],
    'msdf_hash_encode_second_backward': [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_float, C.c_uint32, C.c_int, _P, _P, _P, _P, _P],
    'msdf_hash_scatter_workspace_bytes': [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64],
    'msdf_hash_encode_backward_ws': [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_float, C.c_uint32, C.c_int, _P, _P, C.c_uint64, _P, C.c_uint64, _P],
    'msdf_hash_encode_second_backward_ws': [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_float, C.c_uint32, C.c_int, _P, _P, _P, _P, C.c_uint64, _P,
                                            C.c_uint64, _P],
    'msdf_hash_encode_backward_fused': [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_float, C.c_uint32, _P, C.c_uint64, _P, C.c_uint64, _P],
    'msdf_hash_encode_backward_fused_out': [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_float, C.c_uint32, _P, C.c_uint64, _P, C.c_uint64, _P],
    'msdf_hash_node_forward': [_P, C.c_double, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_float, C.c_uint32, _P, _P],
    'msdf_hash_node_input_gradient': [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, _P, _P],
    'msdf_hash_node_second_grad': [_P, _P, C.c_uint32, C.c_float, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_uint32, _P],
    'msdf_hash_transpose': [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _P],
    'msdf_hash_node_scatter': [_P, _P, C.c_uint32, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                               C.c_uint32, _P, C.c_uint64, _P, C.c_uint64, _P],
    'msdf_weightnorm_forward': [_P, _P, C.c_int, _P, _P, _P, _P],
    'msdf_weightnorm_backward': [_P, _P, C.c_int, _P, _P, _P, _P, _P],
    'msdf_pack_weights': [C.POINTER(Plan), _P, _P, _P, _P, _P, _P, _P],
    'msdf_sdf_forward': [C.POINTER(Plan), _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, _P, _P],
    'msdf_sdf_forward_if': [C.POINTER(Plan), _P, _P, _P, _P, C.c_int, C.c_float, C.c_float, _P, _P, _P],
    'msdf_sdf_forward_lm': [C.POINTER(Plan), _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P, _P],
    'msdf_sdf_fwd_grad': [C.POINTER(Plan), C.POINTER(FgArgs), _P],
    'msdf_sdf_backward': [C.POINTER(Plan), C.POINTER(BwArgs), _P],
    'msdf_color_forward': [C.POINTER(Plan), C.POINTER(ColorFwdArgs), _P],
    'msdf_color_backward': [C.POINTER(Plan), C.POINTER(ColorBwdArgs), _P],
    'msdf_wgrad': [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P],
    'msdf_camera_rays': [_P, _P, _P, C.c_int, _P, _P, _P, _P],
    'msdf_pixel_rays': [_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P],
    'msdf_monosdf_loss': [C.POINTER(MonoSdfLossArgs), _P],
    'msdf_reduce': [_P, C.c_int, _P, _P, _P, _P],
    'msdf_composite_forward': [C.POINTER(CompositeArgs), _P],
    'msdf_composite_backward': [C.POINTER(CompositeBwdArgs), _P],
    'msdf_beta_eff': [_P, C.c_float, _P, _P],
    'msdf_beta_grad': [_P, _P, C.c_int, _P, _P],
    'msdf_probe_loss': [C.POINTER(ProbeLossArgs), _P],
    'msdf_sampler_init': [C.POINTER(SamplerArgs), _P],
    'msdf_sampler_beta': [C.POINTER(SamplerArgs), _P],
    'msdf_sampler_resample': [C.POINTER(SamplerArgs), _P],
    'msdf_sampler_finish': [C.POINTER(SamplerArgs), _P],
    'msdf_sampler_error_bound': [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P],
    'msdf_laplace_density': [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P],
    'msdf_laplace_density_backward': [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P],
}

ABI_VERSION = 8

_lib = None


def load():
    """Load the HIP library once; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'monosdf_amd: %s not found -- build it with `python __graft_entry__.py` or '
            '`make -C monosdf_amd/csrc` (there is no non-HIP fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is missing: intended
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name.endswith('_bytes') else C.c_int
    if lib.msdf_abi_version() != ABI_VERSION:
        raise RuntimeError('monosdf_amd: ABI version mismatch, rebuild the library')
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


# bench.py sets this to a dict to collect (start, end) HIP events per entry point; they are recorded on
# PyTorch's current stream, which is the stream every kernel is launched on (stream_ptr()).
PROFILE = None


# bench.py may restrict the timed entry points to this set (None = all): two events per call are not free
# (~0.2 ms per step for the ~45 calls of a training step), and the roofline needs the large kernels only.
PROFILE_NAMES = None


def call(name, *args):
    prof = PROFILE
    if prof is not None and PROFILE_NAMES is not None and name not in PROFILE_NAMES:
        prof = None
    if prof is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    status = getattr(load(), name)(*args)
    if prof is not None:
        e1.record()
        prof.setdefault(name, []).append((e0, e1))
    if status != 0:
        raise RuntimeError('%s failed: %s' % (name, _ERR.get(status, 'status %d' % status)))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
