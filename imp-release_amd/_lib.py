"""ctypes binding of ``csrc/libimp_hip.so`` (C-ABI declared in ``include/imp_hip.h``).

PyTorch is used only as plumbing here: device memory (``tensor.data_ptr()``) and the current HIP
stream.  There is NO fallback: if the shared library is missing every compute entry point raises
``HipLibraryMissing`` (the product path never routes through ``oracle/`` or torch ops).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMP_HIP_LIB: an alternative build of the same library (A/B runs of kernel variants: tools/build_variant.sh)
LIB_PATH = os.environ.get('IMP_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libimp_hip.so')

MODEL_IDS = {'GM': 0, 'DGNNS': 1, 'AdaGMN': 2}
NORM_IDS = {'in': 0, 'bn': 1}
ACT_IDS = {'relu': 0, 'gelu': 1, 'lrelu': 2}

# every symbol include/imp_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    'imp_last_error', 'imp_version', 'imp_create', 'imp_destroy', 'imp_load_tensor', 'imp_finalize_weights',
    'imp_set_precision', 'imp_get_precision', 'imp_set_sinkhorn_storage', 'imp_num_keys', 'imp_key_name', 'imp_normalize_keypoints', 'imp_encode_keypoints', 'imp_forward_layer',
    'imp_attention_prob', 'imp_attention_received', 'imp_compute_distance', 'imp_compute_score',
    'imp_compute_matches', 'imp_pool', 'imp_score_mass', 'imp_pool_select', 'imp_pool_select_pair', 'imp_masked_commit', 'imp_gather_rows', 'imp_match_pair', 'imp_set_counts', 'imp_match_tail', 'imp_match_tail_scores', 'imp_pool_pair', 'imp_loop_lockstep', 'imp_loop_lockstep_uncertainty', 'imp_op_linear', 'imp_op_layer_gemm', 'imp_op_fused_mlp',
    'imp_op_attention', 'imp_time_attention', 'imp_time_attention_clock', 'imp_time_sinkhorn', 'imp_resident_status', 'imp_resident_health', 'imp_set_resident_verify', 'imp_resident_repaired', 'imp_resident_postmortem', 'imp_ctx_option', 'imp_debug_hold_cus', 'imp_range_events', 'imp_set_range_recovery', 'imp_range_recovered', 'imp_range_take', 'imp_tag_wraps', 'imp_time_layer_gemm', 'imp_estimate_pose', 'imp_pose_stats',
    'imp_sp_create', 'imp_sp_destroy', 'imp_sp_set_weight', 'imp_sp_finalize', 'imp_sp_detect', 'imp_sp_describe', 'imp_sp_dense', 'imp_sp_op_conv',
]


def default_pose_flags():
    """IMP_POSE_MAGSAC, + IMP_POSE_ADAPTIVE with IMP_POSE_ADAPTIVE=1 in the environment (include/imp_hip.h).  Adaptive termination is OFF by
    default: measured in the configs[4] loops (profiles/r05/pose_adaptive_*.log) it LOWERS the rate - 728-742 vs 804-807 pairs/s (IMP),
    797-831 vs 853-900 (EIMP) - because most estimates of the harder set need more than the first 128 samples and then pay a second
    solver round (+0.1 ms of latency per estimate, 0.33 vs 0.19 ms per call) while the fixed budget's 1024 samples run side by side in one"""
    return 1 | (4 if os.environ.get('IMP_POSE_ADAPTIVE', '0') == '1' else 0)


class HipLibraryMissing(RuntimeError):
    pass


class ImpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libimp_hip error {code}: {msg}')
        self.code = code


class ResidentSinkhornTimeout(ImpError):
    """IMP_E_RESIDENT: a chip-resident Sinkhorn launch of an EARLIER call on this context was voided (its outputs are poisoned:
    mscores NaN, indices -1).  The context has already recovered on a safer protocol; re-run the batch."""


class OperandRangeError(ImpError):
    """IMP_E_RANGE: an EARLIER call on this context produced non-finite match scores - in the default f16x3 arithmetic an operand
    left the fp16 range (|x| >= 65504), or the inputs were not finite.  That call's matches are void (all -1); use
    ``precision='f32'`` for such data."""


class ResidentDoesNotFit(ImpError):
    """IMP_E_NOFIT: a RAGGED batch (set_counts) that the chip-resident Sinkhorn kernel cannot hold; nothing was computed - run the pairs in
    smaller groups (a single pair always runs).  Not a time-out: :class:`ResidentSinkhornTimeout` is a different class on purpose."""


IMP_E_RESIDENT = -6
IMP_E_RANGE = -7
IMP_E_NOFIT = -8


class ImpLoopPair(C.Structure):
    """include/imp_hip.h imp_loop_pair"""
    _fields_ = [('pts0', C.c_void_p), ('pts1', C.c_void_p), ('K0', C.c_void_p), ('K1', C.c_void_p), ('indices0', C.c_void_p), ('mscores0', C.c_void_p),
                ('R', C.c_double * 9), ('t', C.c_double * 3), ('found', C.c_int32), ('n_iterations', C.c_int32)]


class ImpLoopPairU(C.Structure):
    """include/imp_hip.h imp_loop_pair_u"""
    _fields_ = [('pts0', C.c_void_p), ('pts1', C.c_void_p), ('K0', C.c_void_p), ('K1', C.c_void_p), ('indices0', C.c_void_p), ('mscores0', C.c_void_p),
                ('kept0', C.c_void_p), ('kept1', C.c_void_p), ('R', C.c_double * 9), ('t', C.c_double * 3), ('found', C.c_int32),
                ('n_iterations', C.c_int32), ('n_kept0', C.c_int32), ('n_kept1', C.c_int32), ('n_indices', C.c_int32), ('reserved', C.c_int32)]


class ImpConfig(C.Structure):
    _fields_ = [('model', C.c_int32), ('descriptor_dim', C.c_int32), ('n_gnn_layers', C.c_int32),
                ('n_layers', C.c_int32), ('kenc_channels', C.c_int32 * 8), ('norm_fn', C.c_int32),
                ('ac_fn', C.c_int32), ('max_batch', C.c_int32), ('max_keypoints', C.c_int32),
                ('layer_is_cross', C.c_int32 * 128)]


_lib = None


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(['make', '-C', os.path.join(_HERE, 'csrc'), '-j8'] + (['-B'] if force else []), check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C imp-release_amd/csrc`).  There is no CPU fallback for the matching hot path.')
    L = C.CDLL(LIB_PATH)
    L.imp_last_error.restype = C.c_char_p
    L.imp_version.restype = C.c_char_p
    L.imp_key_name.restype = C.c_char_p
    L.imp_key_name.argtypes = [C.c_void_p, C.c_int]
    L.imp_num_keys.argtypes = [C.c_void_p]
    L.imp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(ImpConfig), C.c_int]
    L.imp_destroy.argtypes = [C.c_void_p]
    L.imp_load_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    L.imp_finalize_weights.argtypes = [C.c_void_p]
    L.imp_set_precision.argtypes = [C.c_void_p, C.c_int]
    L.imp_get_precision.argtypes = [C.c_void_p]
    L.imp_set_sinkhorn_storage.argtypes = [C.c_void_p, C.c_int]
    P, I, F = C.c_void_p, C.c_int, C.c_float
    L.imp_normalize_keypoints.argtypes = [P, P, I, I, F, F, P, P]
    L.imp_encode_keypoints.argtypes = [P, I, I, I, P, P, P, P, P, P, P, P, P]
    L.imp_forward_layer.argtypes = [P, I, I, I, I, P, P, P, P, P, P, P]
    L.imp_attention_prob.argtypes = [P, I, P, P]
    L.imp_attention_received.argtypes = [P, I, P, P]
    L.imp_compute_distance.argtypes = [P, I, I, I, I, P, P, P, P]
    L.imp_compute_score.argtypes = [P, I, I, I, P, F, I, I, P, P]
    L.imp_compute_matches.argtypes = [P, I, I, I, P, F, P, P, P, P, P]
    L.imp_pool.argtypes = [P, I, I, P, F, F, I, P, P, P, P]
    L.imp_score_mass.argtypes = [P, I, I, P, P, P, P]
    L.imp_pool_select.argtypes = [P, I, P, P, P, F, P, P, P]
    L.imp_pool_select_pair.argtypes = [P, I, P, P, P, I, P, I, P, P, P, I, P, F, P, P]
    L.imp_gather_rows.argtypes = [P, I, I, I, I, P, P, P, P]
    L.imp_masked_commit.argtypes = [P, I, P, P, P, P, P, P, P, I, P, I, P, P, P, P, P]
    L.imp_match_pair.argtypes = [P, I, I, I, P, P, P, P, P, P, F, F, F, I, I, F, P, P, P, P, P, P]
    L.imp_set_counts.argtypes = [P, I, P, P]
    L.imp_match_tail.argtypes = [P, I, I, I, I, P, P, F, I, I, F, P, P, P, P, P]
    L.imp_match_tail_scores.argtypes = [P, I, I, I, I, P, P, F, I, I, F, P, P, P, P, P, P]
    L.imp_pool_pair.argtypes = [P, I, I, I, I, P, F, F, I, P, P, P, P]
    L.imp_loop_lockstep.argtypes = [P, I, P, P, I, I, P, P, P, P, P, P, F, I, I, C.c_uint, F, I, C.c_double, C.c_double, I, I, C.c_uint, I, P, P]
    L.imp_loop_lockstep_uncertainty.argtypes = [P, I, P, P, I, I, P, P, P, P, P, P, F, I, I, C.c_uint, F, I, C.c_double, C.c_double, I, I, I, I, C.c_uint, I, P, P]
    L.imp_op_linear.argtypes = [P, I, I, I, P, P, P, P, P]
    L.imp_op_layer_gemm.argtypes = [P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, I, P, I, P]
    L.imp_op_fused_mlp.argtypes = [P, I, I, P, P, P, P, P, P, P, P, I, P, P, I, P]
    L.imp_op_attention.argtypes = [P, I, I, I, I, P, P, P, P, P, P]
    L.imp_time_attention.argtypes = [P, I, I, I, C.POINTER(C.c_float), P]
    L.imp_time_attention_clock.argtypes = [P, I, I, I, C.POINTER(C.c_float), C.POINTER(C.c_float), P]
    L.imp_time_sinkhorn.argtypes = [P, I, I, I, C.POINTER(C.c_float), P]
    L.imp_resident_status.argtypes = [P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.imp_resident_health.argtypes = [P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.imp_set_resident_verify.argtypes = [P, I]
    L.imp_resident_repaired.argtypes = [P]
    L.imp_debug_hold_cus.argtypes = [I, I, I, P]
    L.imp_resident_postmortem.argtypes = [P, C.POINTER(C.c_int32), I]
    L.imp_range_events.argtypes = [P]
    L.imp_set_range_recovery.argtypes = [P, I]
    L.imp_range_recovered.argtypes = [P]
    L.imp_range_take.argtypes = [P, I]
    L.imp_tag_wraps.argtypes = [P]
    L.imp_time_layer_gemm.argtypes = [P, I, I, I, I, I, C.POINTER(C.c_float), P]
    L.imp_estimate_pose.argtypes = [P, P, I, P, P, C.c_double, I, C.c_uint, I, P, P, P, P, P, C.POINTER(C.c_int), I, P]
    L.imp_sp_create.argtypes = [C.POINTER(C.c_void_p), I, I]
    L.imp_sp_destroy.argtypes = [P]
    L.imp_sp_destroy.restype = None
    L.imp_sp_set_weight.argtypes = [P, C.c_char_p, P, C.c_int64]
    L.imp_sp_finalize.argtypes = [P]
    L.imp_sp_detect.argtypes = [P, P, I, I, I, I, F, I, I, I, P, C.POINTER(C.c_int)]
    L.imp_sp_describe.argtypes = [P, I, P, P, P, P]
    L.imp_sp_dense.argtypes = [P, P, P, P, P]
    L.imp_sp_op_conv.argtypes = [P, I, P, I, I, I, P, I, I, P]
    _lib = L
    return L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device=None):
    """the calling thread's current stream ON `device` (not on the thread's current device: worker threads of a
    multi-GPU rank start with device 0 current)"""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f'{name} must live on the GPU (got {t.device}); libimp_hip has no CPU path')
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class Context:
    """Owns one ``imp_ctx`` (packed weights + workspace) on one device."""

    def __init__(self, config: dict, model: str, device: torch.device, max_batch: int = 0, max_keypoints: int = 0):
        self.L = lib()
        cfg = ImpConfig()
        cfg.model = MODEL_IDS[model]
        cfg.descriptor_dim = int(config['descriptor_dim'])
        names = list(config['GNN_layers'])
        cfg.n_gnn_layers = len(names)
        cfg.n_layers = int(config['n_layers'])
        kc = list(config['keypoint_encoder'])
        if len(kc) > 7:
            raise ValueError('keypoint_encoder: at most 7 hidden layers')
        for i, v in enumerate(kc):
            cfg.kenc_channels[i] = int(v)
        if config['norm_fn'] not in NORM_IDS or config['ac_fn'] not in ACT_IDS:
            raise ValueError("norm_fn must be 'in'|'bn' and ac_fn 'relu'|'gelu'|'lrelu'")
        cfg.norm_fn = NORM_IDS[config['norm_fn']]
        cfg.ac_fn = ACT_IDS[config['ac_fn']]
        cfg.max_batch = max_batch
        cfg.max_keypoints = max_keypoints
        for i, nm in enumerate(names):
            if nm not in ('self', 'cross'):
                raise ValueError(f'unknown GNN layer name {nm!r}')
            cfg.layer_is_cross[i] = 1 if nm == 'cross' else 0
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('imp_release_amd needs a GPU device; there is no CPU path')
        self.handle = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device('cuda', idx)
        self._check(self.L.imp_create(C.byref(self.handle), C.byref(cfg), idx))
        self.D = cfg.descriptor_dim
        prec = config.get('precision')          # extra (non-reference) config key: 'f16x3' (default) | 'f32'
        if prec is not None:
            if prec not in ('f32', 'f16x3'):
                raise ValueError("precision must be 'f32' or 'f16x3'")
            self._check(self.L.imp_set_precision(self.handle, 1 if prec == 'f16x3' else 0))
        stor = config.get('sinkhorn_storage')   # extra config key: 4 (default, fp32) | 3 (3-byte copy for the iterations)
        if stor is not None:
            self._check(self.L.imp_set_sinkhorn_storage(self.handle, int(stor)))
        # extra config key 'range_recovery' (default True; env IMP_RANGE_RECOVERY=0 turns the default off): the one-shot / tail calls wait for
        # their own work and re-run a call whose operands left the fp16 range on the fp32 path (include/imp_hip.h imp_set_range_recovery)
        rr = config.get('range_recovery')
        if rr is None:
            rr = os.environ.get('IMP_RANGE_RECOVERY', '1') != '0'
        self._check(self.L.imp_set_range_recovery(self.handle, 1 if rr else 0))
        self.range_recovery = bool(rr)

    def set_range_recovery(self, on: bool):
        self._check(self.L.imp_set_range_recovery(self.handle, 1 if on else 0))
        self.range_recovery = bool(on)

    def set_precision(self, name: str):
        self._check(self.L.imp_set_precision(self.handle, 1 if name == 'f16x3' else 0))

    def range_take(self, recovered: bool) -> bool:
        """after the caller synchronised: did a match kernel of the pass meet non-finite scores?  Clears and counts the event (imp_range_take)"""
        return bool(self.L.imp_range_take(self.handle, 1 if recovered else 0))

    def range_counts(self):
        """(calls that met non-finite scores, calls repaired in place on the fp32 path)"""
        return self.L.imp_range_events(self.handle), self.L.imp_range_recovered(self.handle)

    def set_sinkhorn_storage(self, bytes_per_element: int):
        """4 = the Sinkhorn iterations stream the fp32 matrix (default), 3 = the 3-byte copy (include/imp_hip.h)"""
        self._check(self.L.imp_set_sinkhorn_storage(self.handle, int(bytes_per_element)))

    @property
    def precision(self):
        return 'f16x3' if self.L.imp_get_precision(self.handle) == 1 else 'f32'

    def _check(self, rc):
        if rc != 0:
            cls = {IMP_E_RESIDENT: ResidentSinkhornTimeout, IMP_E_RANGE: OperandRangeError, IMP_E_NOFIT: ResidentDoesNotFit}.get(rc, ImpError)
            raise cls(rc, self.L.imp_last_error().decode())

    def close(self):
        if getattr(self, 'handle', None) and self.handle.value:
            self.L.imp_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ------------------------------------------------------------------------------
    def keys(self):
        return [self.L.imp_key_name(self.handle, i).decode() for i in range(self.L.imp_num_keys(self.handle))]

    def load_state_dict(self, sd):
        want = set(self.keys())
        for k, v in sd.items():
            if k not in want:
                if k.endswith('num_batches_tracked'):
                    continue
                raise KeyError(f'unexpected state_dict key {k}')
            t = torch.as_tensor(v).detach().to('cpu', torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            self._check(self.L.imp_load_tensor(self.handle, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
        self._check(self.L.imp_finalize_weights(self.handle))

    # -- ops ----------------------------------------------------------------------------------
    def normalize_keypoints(self, kpts, width, height):
        kpts = _f32(kpts, 'kpts')
        out = torch.empty_like(kpts)
        B, n = kpts.shape[0], kpts.shape[1]
        self._check(self.L.imp_normalize_keypoints(self.handle, _ptr(kpts), B, n, float(width), float(height),
                                                   _ptr(out), _stream(self.device)))
        return out

    def encode_keypoints(self, nk0, sc0, nk1, sc1, desc0=None, desc1=None):
        nk0, sc0, nk1, sc1 = _f32(nk0, 'kpts0'), _f32(sc0, 'scores0'), _f32(nk1, 'kpts1'), _f32(sc1, 'scores1')
        B, n0, n1 = nk0.shape[0], nk0.shape[1], nk1.shape[1]
        d0 = None if desc0 is None else _f32(desc0, 'desc0')
        d1 = None if desc1 is None else _f32(desc1, 'desc1')
        o0 = torch.empty(B, n0, self.D, device=nk0.device, dtype=torch.float32)
        o1 = torch.empty(B, n1, self.D, device=nk0.device, dtype=torch.float32)
        self._check(self.L.imp_encode_keypoints(self.handle, B, n0, n1, _ptr(nk0), _ptr(sc0), _ptr(d0), _ptr(o0),
                                                _ptr(nk1), _ptr(sc1), _ptr(d1), _ptr(o1), _stream(self.device)))
        return o0, o1

    def forward_layer(self, layer_i, desc0, desc1, mask0=None, mask1=None, inplace=False):
        desc0, desc1 = _f32(desc0, 'desc0'), _f32(desc1, 'desc1')
        B, n0, n1 = desc0.shape[0], desc0.shape[1], desc1.shape[1]
        o0 = desc0 if inplace else torch.empty_like(desc0)
        o1 = desc1 if inplace else torch.empty_like(desc1)
        if mask0 is not None:
            mask0 = mask0.to(torch.uint8).contiguous()
        if mask1 is not None:
            mask1 = mask1.to(torch.uint8).contiguous()
        self._check(self.L.imp_forward_layer(self.handle, layer_i, B, n0, n1, _ptr(desc0), _ptr(desc1), _ptr(o0),
                                             _ptr(o1), _ptr(mask0), _ptr(mask1), _stream(self.device)))
        return o0, o1

    def attention_prob(self, which, B, nq, nk, device):
        out = torch.empty(B, 4, nq, nk, device=device, dtype=torch.float32)
        self._check(self.L.imp_attention_prob(self.handle, which, _ptr(out), _stream(self.device)))
        return out

    def attention_received(self, which, B, nk, device):
        out = torch.empty(B, nk, device=device, dtype=torch.float32)
        self._check(self.L.imp_attention_received(self.handle, which, _ptr(out), _stream(self.device)))
        return out

    def compute_distance(self, layer_id, desc0, desc1):
        desc0, desc1 = _f32(desc0, 'desc0'), _f32(desc1, 'desc1')
        B, n0, n1 = desc0.shape[0], desc0.shape[1], desc1.shape[1]
        dist = torch.empty(B, n0, n1, device=desc0.device, dtype=torch.float32)
        self._check(self.L.imp_compute_distance(self.handle, layer_id, B, n0, n1, _ptr(desc0), _ptr(desc1),
                                                _ptr(dist), _stream(self.device)))
        return dist

    def compute_score(self, dist, bin_score, iterations, with_sinkhorn=True):
        dist = _f32(dist, 'dist')
        B, n0, n1 = dist.shape
        scores = torch.empty(B, n0 + 1, n1 + 1, device=dist.device, dtype=torch.float32)
        self._check(self.L.imp_compute_score(self.handle, B, n0, n1, _ptr(dist), float(bin_score), int(iterations),
                                             1 if with_sinkhorn else 0, _ptr(scores), _stream(self.device)))
        return scores

    def compute_matches(self, scores, p):
        scores = _f32(scores, 'scores')
        B, n0, n1 = scores.shape[0], scores.shape[1] - 1, scores.shape[2] - 1
        dev = scores.device
        i0 = torch.empty(B, n0, device=dev, dtype=torch.int64)
        i1 = torch.empty(B, n1, device=dev, dtype=torch.int64)
        m0 = torch.empty(B, n0, device=dev, dtype=torch.float32)
        m1 = torch.empty(B, n1, device=dev, dtype=torch.float32)
        self._check(self.L.imp_compute_matches(self.handle, B, n0, n1, _ptr(scores), float(p), _ptr(i0), _ptr(i1),
                                               _ptr(m0), _ptr(m1), _stream(self.device)))
        return i0, i1, m0, m1

    def pool(self, scores, mscore_th, uncertainty_ratio, n_min_tokens, return_host=False):
        """-> (ids0 | None, ids1 | None); ONE host sync: the kept counts AND the id lists come back in a single copy
        (shapes change, and the loop needs the ids on the host anyway to slice its CPU keypoints).
        ``return_host``: additionally return the two id lists as numpy arrays."""
        scores = _f32(scores, 'scores')
        n0, n1 = scores.shape[1] - 1, scores.shape[2] - 1
        dev = scores.device
        buf = torch.empty(n0 + n1 + 2, device=dev, dtype=torch.int64)          # ids0 | ids1 | 4 int32 counts
        ids0, ids1 = buf[:n0], buf[n0:n0 + n1]
        counts = buf[n0 + n1:].view(torch.int32)
        self._check(self.L.imp_pool(self.handle, n0, n1, _ptr(scores), float(mscore_th), float(uncertainty_ratio),
                                    int(n_min_tokens), _ptr(ids0), _ptr(ids1), _ptr(counts), _stream(self.device)))
        host = buf.cpu()
        c = host[n0 + n1:].view(torch.int32).tolist()
        r0 = ids0[:c[0]] if c[0] >= 0 else None
        r1 = ids1[:c[2]] if c[2] >= 0 else None
        if not return_host:
            return r0, r1
        h0 = host[:c[0]].numpy() if c[0] >= 0 else None
        h1 = host[n0:n0 + c[2]].numpy() if c[2] >= 0 else None
        return r0, r1, h0, h1

    def score_mass(self, scores):
        scores = _f32(scores, 'scores')
        n0, n1 = scores.shape[-2] - 1, scores.shape[-1] - 1
        m0 = torch.empty(n0, device=scores.device, dtype=torch.float32)
        m1 = torch.empty(n1, device=scores.device, dtype=torch.float32)
        self._check(self.L.imp_score_mass(self.handle, n0, n1, _ptr(scores), _ptr(m0), _ptr(m1), _stream(self.device)))
        return m0, m1

    def pool_select(self, mass, a_self, a_cross, thr):
        mass, a_self, a_cross = _f32(mass, 'mass'), _f32(a_self, 'a_self'), _f32(a_cross, 'a_cross')
        n = mass.numel()
        ids = torch.empty(n, device=mass.device, dtype=torch.int64)
        counts = torch.empty(2, device=mass.device, dtype=torch.int32)
        self._check(self.L.imp_pool_select(self.handle, n, _ptr(mass), _ptr(a_self), _ptr(a_cross), float(thr),
                                           _ptr(ids), _ptr(counts), _stream(self.device)))
        c = counts.tolist()
        return ids[:c[0]] if c[0] >= 0 else None

    def pool_select_pair(self, mass0, a_self0, a_cross0, skip0, mass1, a_self1, a_cross1, skip1, thr):
        """both images of a pair in one launch and ONE count read-back: -> (ids0 | None, ids1 | None) (None: side skipped or nothing confident)"""
        v = [_f32(t, 'pool vector') for t in (mass0, a_self0, a_cross0, mass1, a_self1, a_cross1)]
        n0, n1 = v[0].numel(), v[3].numel()
        dev = v[0].device
        ids0 = torch.empty(n0, device=dev, dtype=torch.int64)
        ids1 = torch.empty(n1, device=dev, dtype=torch.int64)
        counts = torch.empty(4, device=dev, dtype=torch.int32)
        self._check(self.L.imp_pool_select_pair(self.handle, n0, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), 1 if skip0 else 0, _ptr(ids0),
                                                n1, _ptr(v[3]), _ptr(v[4]), _ptr(v[5]), 1 if skip1 else 0, _ptr(ids1), float(thr),
                                                _ptr(counts), _stream(self.device)))
        c = counts.tolist()
        return (ids0[:c[0]] if c[0] >= 0 else None), (ids1[:c[2]] if c[2] >= 0 else None)

    def masked_commit(self, g0, g1, i0, m0, out_i_row, out_m_row, keep0=None, keep1=None, mask0_row=None, mask1_row=None, update=False):
        """masked AdaGMN bookkeeping of one pair in one launch (include/imp_hip.h imp_masked_commit): scatters the matches of the kept
        keypoints into the pair's full-size rows; with ``update`` also returns the id lists composed with the pool's selection
        (``keep_s`` None: unchanged) after entering them into the mask rows.  No synchronisation."""
        n0sel = g0.numel()
        ng0 = ng1 = None
        nk0 = nk1 = 0
        if update:
            nk0 = keep0.numel() if keep0 is not None else g0.numel()
            nk1 = keep1.numel() if keep1 is not None else g1.numel()
            ng0 = torch.empty(nk0, device=g0.device, dtype=torch.int64)
            ng1 = torch.empty(nk1, device=g1.device, dtype=torch.int64)
        N = lambda t: _ptr(t) if t is not None else None      # noqa: E731
        self._check(self.L.imp_masked_commit(self.handle, n0sel, _ptr(g0), _ptr(g1), _ptr(i0), _ptr(m0), _ptr(out_i_row), _ptr(out_m_row),
                                             N(keep0), nk0, N(keep1), nk1, N(ng0), N(ng1), N(mask0_row if update else None),
                                             N(mask1_row if update else None), _stream(self.device)))
        return ng0, ng1

    def gather_rows(self, x, ids, out=None):
        """x [B, n_in, dim], ids [n_out] -> [B, n_out, dim]; ``out``: a contiguous float32 tensor whose first B * n_out * dim elements
        receive the rows (a pair's slot of a padded batch) instead of a new tensor"""
        x = _f32(x, 'x')
        ids = ids.to(torch.int64).contiguous()
        B, n_in, dim = x.shape
        if out is None:
            out = torch.empty(B, ids.numel(), dim, device=x.device, dtype=torch.float32)
        elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < B * ids.numel() * dim:
            raise ValueError('gather_rows: out must be a contiguous float32 tensor with room for the gathered rows')
        self._check(self.L.imp_gather_rows(self.handle, B, n_in, ids.numel(), dim, _ptr(x), _ptr(ids), _ptr(out),
                                           _stream(self.device)))
        return out

    def match_pair(self, kpts0, sc0, desc0, kpts1, sc1, desc1, width, height, bin_score, iterations, with_sinkhorn,
                   p, want_scores=False, want_side1=False, out=None):
        kpts0, sc0, desc0 = _f32(kpts0, 'kpts0'), _f32(sc0, 'scores0'), _f32(desc0, 'desc0')
        kpts1, sc1, desc1 = _f32(kpts1, 'kpts1'), _f32(sc1, 'scores1'), _f32(desc1, 'desc1')
        B, n0, n1 = kpts0.shape[0], kpts0.shape[1], kpts1.shape[1]
        dev = kpts0.device
        if out is None:
            out = {'indices0': torch.empty(B, n0, device=dev, dtype=torch.int64),
                   'mscores0': torch.empty(B, n0, device=dev, dtype=torch.float32)}
            if want_side1:
                out['indices1'] = torch.empty(B, n1, device=dev, dtype=torch.int64)
                out['mscores1'] = torch.empty(B, n1, device=dev, dtype=torch.float32)
            if want_scores:
                out['scores'] = torch.empty(B, n0 + 1, n1 + 1, device=dev, dtype=torch.float32)
        self._check(self.L.imp_match_pair(
            self.handle, B, n0, n1, _ptr(kpts0), _ptr(sc0), _ptr(desc0), _ptr(kpts1), _ptr(sc1), _ptr(desc1),
            float(width), float(height), float(bin_score), int(iterations), 1 if with_sinkhorn else 0, float(p),
            _ptr(out['indices0']), _ptr(out['mscores0']), _ptr(out.get('indices1')), _ptr(out.get('mscores1')),
            _ptr(out.get('scores')), _stream(self.device)))
        return out

    def set_counts(self, n0=None, n1=None):
        """ragged batches (include/imp_hip.h imp_set_counts): per-pair keypoint counts (host sequences) that the NEXT calls on this
        context obey - tensors stay padded to the largest pair; ``set_counts()`` returns to uniform batches"""
        if n0 is None and n1 is None:
            self._check(self.L.imp_set_counts(self.handle, 0, None, None))
            return
        a0 = (C.c_int32 * len(n0))(*[int(v) for v in n0])
        a1 = (C.c_int32 * len(n1))(*[int(v) for v in n1])
        if len(n0) != len(n1):
            raise ValueError('set_counts: one count per pair and image')
        self._check(self.L.imp_set_counts(self.handle, len(n0), a0, a1))

    def match_tail(self, layer_id, desc0, desc1, bin_score, iterations, with_sinkhorn, p, want_side1=False, want_scores=False):
        """final projection of iteration `layer_id` -> distance -> Sinkhorn -> mutual matches without a score tensor
        (include/imp_hip.h imp_match_tail); obeys set_counts.  ``want_scores``: additionally ``out['scores']``, [B, (n0 + 1) * (n1 + 1)]
        - slot b starts with pair b's DENSE score tensor of its own shape (imp_match_tail_scores; what the pool consumes)"""
        desc0, desc1 = _f32(desc0, 'desc0'), _f32(desc1, 'desc1')
        B, n0, n1 = desc0.shape[0], desc0.shape[1], desc1.shape[1]
        dev = desc0.device
        out = {'indices0': torch.empty(B, n0, device=dev, dtype=torch.int64), 'mscores0': torch.empty(B, n0, device=dev, dtype=torch.float32)}
        if want_side1:
            out['indices1'] = torch.empty(B, n1, device=dev, dtype=torch.int64)
            out['mscores1'] = torch.empty(B, n1, device=dev, dtype=torch.float32)
        if want_scores:
            out['scores'] = torch.empty(B, (n0 + 1) * (n1 + 1), device=dev, dtype=torch.float32)
        self._check(self.L.imp_match_tail_scores(self.handle, int(layer_id), B, n0, n1, _ptr(desc0), _ptr(desc1), float(bin_score), int(iterations),
                                                 1 if with_sinkhorn else 0, float(p), _ptr(out['indices0']), _ptr(out['mscores0']),
                                                 _ptr(out.get('indices1')), _ptr(out.get('mscores1')), _ptr(out.get('scores')), _stream(self.device)))
        return out

    def pool_pairs(self, jobs, batch, n0, n1, scores, uncertainty_ratio, n_min_tokens):
        """AdaGMN.pool of several pairs of the batch whose attention is cached (include/imp_hip.h imp_pool_pair), ONE host sync for all:
        ``jobs`` = [(pair, (m0, m1), mscore_th)]; ``scores`` = the [batch, (n0 + 1) * (n1 + 1)] tensor of match_tail(want_scores=True).
        -> {pair: (ids0 | None, ids1 | None, host ids0 | None, host ids1 | None)} like ``pool(..., return_host=True)``"""
        if not jobs:
            return {}
        dev = scores.device
        w = n0 + n1 + 2
        buf = torch.empty(len(jobs), w, device=dev, dtype=torch.int64)          # per pair: ids0 | ids1 | 4 int32 counts
        for k, (b, (m0, m1), th) in enumerate(jobs):
            row = buf[k]
            self._check(self.L.imp_pool_pair(self.handle, int(b), int(batch), int(n0), int(n1), _ptr(scores[b]), float(th), float(uncertainty_ratio),
                                             int(n_min_tokens), _ptr(row[:n0]), _ptr(row[n0:n0 + n1]), _ptr(row[n0 + n1:]), _stream(self.device)))
        host = buf.cpu()
        out = {}
        for k, (b, _, _) in enumerate(jobs):
            c = host[k, n0 + n1:].view(torch.int32).tolist()
            out[b] = (buf[k, :c[0]] if c[0] >= 0 else None, buf[k, n0:n0 + c[2]] if c[2] >= 0 else None,
                      host[k, :c[0]].numpy() if c[0] >= 0 else None, host[k, n0:n0 + c[2]].numpy() if c[2] >= 0 else None)
        return out

    def loop_lockstep(self, n0, n1, nk0, sc0, de0, nk1, sc1, de1, pts0, pts1, K0, K1, bin_score, sinkhorn_iterations, n_iterations, valid_its,
                      match_ratio, min_kpts, error_th, stop_pose_deg, pose_threads=4, pose_iterations=1024, pose_seed=1, pose_flags=None):
        """the native lock-step IMP loop (include/imp_hip.h imp_loop_lockstep): padded device tensors + per-pair host arrays in,
        [(indices0, mscores0, R | None, t | None, n_iterations)] out"""
        import numpy as np
        B = len(n0)
        nk0, sc0, de0, nk1, sc1, de1 = (_f32(t, 'input') for t in (nk0, sc0, de0, nk1, sc1, de1))
        N0, N1 = nk0.shape[1], nk1.shape[1]
        recs = (ImpLoopPair * B)()
        keep = []
        for b in range(B):
            p0 = np.ascontiguousarray(pts0[b], dtype=np.float32); p1 = np.ascontiguousarray(pts1[b], dtype=np.float32)
            k0 = np.ascontiguousarray(np.eye(3) if K0[b] is None else K0[b], dtype=np.float64).reshape(3, 3)
            k1 = np.ascontiguousarray(np.eye(3) if K1[b] is None else K1[b], dtype=np.float64).reshape(3, 3)
            oi = np.empty(n0[b], dtype=np.int64); om = np.empty(n0[b], dtype=np.float32)
            keep.append((p0, p1, k0, k1, oi, om))
            recs[b].pts0, recs[b].pts1 = p0.ctypes.data, p1.ctypes.data
            recs[b].K0, recs[b].K1 = k0.ctypes.data, k1.ctypes.data
            recs[b].indices0, recs[b].mscores0 = oi.ctypes.data, om.ctypes.data
        a0 = (C.c_int32 * B)(*[int(v) for v in n0]); a1 = (C.c_int32 * B)(*[int(v) for v in n1])
        mask = 0
        for it in valid_its:
            mask |= 1 << int(it)
        self._check(self.L.imp_loop_lockstep(self.handle, B, a0, a1, N0, N1, _ptr(nk0), _ptr(sc0), _ptr(de0), _ptr(nk1), _ptr(sc1), _ptr(de1),
                                             float(bin_score), int(sinkhorn_iterations), int(n_iterations), C.c_uint(mask), float(match_ratio), int(min_kpts),
                                             C.c_double(float(error_th)), C.c_double(float(stop_pose_deg)), int(pose_threads), int(pose_iterations),
                                             C.c_uint(pose_seed), int(default_pose_flags() if pose_flags is None else pose_flags), recs, _stream(self.device)))
        out = []
        for b in range(B):
            found = bool(recs[b].found)
            R = np.array(recs[b].R, dtype=np.float64).reshape(3, 3) if found else None
            t = np.array(recs[b].t, dtype=np.float64) if found else None
            out.append((keep[b][4], keep[b][5], R, t, int(recs[b].n_iterations)))
        return out

    def loop_lockstep_uncertainty(self, n0, n1, nk0, sc0, de0, nk1, sc1, de1, pts0, pts1, K0, K1, bin_score, sinkhorn_iterations, n_iterations,
                                  valid_its, match_ratio, min_kpts, error_th, stop_pose_deg, with_uncertainty, n_min_tokens=256, pose_threads=4,
                                  pose_iterations=1024, pose_seed=1, pose_flags=None):
        """the native lock-step EIMP loop (include/imp_hip.h imp_loop_lockstep_uncertainty): padded device tensors + per-pair host arrays
        in, [(kept0, kept1, indices0, mscores0, R | None, t | None, n_iterations)] out (kept: indices of the surviving keypoints)"""
        import numpy as np
        B = len(n0)
        nk0, sc0, de0, nk1, sc1, de1 = (_f32(t, 'input') for t in (nk0, sc0, de0, nk1, sc1, de1))
        N0, N1 = nk0.shape[1], nk1.shape[1]
        recs = (ImpLoopPairU * B)()
        keep = []
        for b in range(B):
            p0 = np.ascontiguousarray(pts0[b], dtype=np.float32); p1 = np.ascontiguousarray(pts1[b], dtype=np.float32)
            k0 = np.ascontiguousarray(np.eye(3) if K0[b] is None else K0[b], dtype=np.float64).reshape(3, 3)
            k1 = np.ascontiguousarray(np.eye(3) if K1[b] is None else K1[b], dtype=np.float64).reshape(3, 3)
            oi = np.empty(n0[b], dtype=np.int64); om = np.empty(n0[b], dtype=np.float32)
            q0 = np.empty(n0[b], dtype=np.int32); q1 = np.empty(n1[b], dtype=np.int32)
            keep.append((p0, p1, k0, k1, oi, om, q0, q1))
            recs[b].pts0, recs[b].pts1 = p0.ctypes.data, p1.ctypes.data
            recs[b].K0, recs[b].K1 = k0.ctypes.data, k1.ctypes.data
            recs[b].indices0, recs[b].mscores0 = oi.ctypes.data, om.ctypes.data
            recs[b].kept0, recs[b].kept1 = q0.ctypes.data, q1.ctypes.data
        a0 = (C.c_int32 * B)(*[int(v) for v in n0]); a1 = (C.c_int32 * B)(*[int(v) for v in n1])
        mask = 0
        for it in valid_its:
            mask |= 1 << int(it)
        self._check(self.L.imp_loop_lockstep_uncertainty(
            self.handle, B, a0, a1, N0, N1, _ptr(nk0), _ptr(sc0), _ptr(de0), _ptr(nk1), _ptr(sc1), _ptr(de1), float(bin_score), int(sinkhorn_iterations),
            int(n_iterations), C.c_uint(mask), float(match_ratio), int(min_kpts), C.c_double(float(error_th)), C.c_double(float(stop_pose_deg)),
            1 if with_uncertainty else 0, int(n_min_tokens), int(pose_threads), int(pose_iterations), C.c_uint(pose_seed), int(default_pose_flags() if pose_flags is None else pose_flags), recs,
            _stream(self.device)))
        out = []
        for b in range(B):
            r = recs[b]
            found = bool(r.found)
            R = np.array(r.R, dtype=np.float64).reshape(3, 3) if found else None
            t = np.array(r.t, dtype=np.float64) if found else None
            ni = int(r.n_indices)
            out.append((keep[b][6][:int(r.n_kept0)].astype(np.int64), keep[b][7][:int(r.n_kept1)].astype(np.int64), keep[b][4][:ni], keep[b][5][:ni], R, t,
                        int(r.n_iterations)))
        return out

    def op_linear(self, x, W, bias=None):
        x, W = _f32(x, 'x'), _f32(W, 'W')
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        b = None if bias is None else _f32(bias, 'bias')
        self._check(self.L.imp_op_linear(self.handle, M, N, K, _ptr(x), _ptr(W), _ptr(b), _ptr(y), _stream(self.device)))
        return y

    def op_layer_gemm(self, x, W, bias=None, x2=None, residual=None, stats_in=None, want_stats=False, W2=None, bias2=None, pass_split=1):
        """csrc/gemm_wf.hip on its own (include/imp_hip.h imp_op_layer_gemm): x [B, M, ksplit] (+ x2 [B, M, K - ksplit]) -> y [B, M, N]
        (+ (mean, rstd) [B, N, 2] of y, + the chained y2 [B, M, N2])"""
        x, W = _f32(x, 'x'), _f32(W, 'W')
        B, M, ks = x.shape
        N, K = W.shape
        dev = x.device
        opt = lambda t, n: None if t is None else _f32(t, n)
        x2, bias, residual, stats_in, W2, bias2 = opt(x2, 'x2'), opt(bias, 'bias'), opt(residual, 'residual'), opt(stats_in, 'stats_in'), opt(W2, 'W2'), opt(bias2, 'bias2')
        y = torch.empty(B, M, N, device=dev, dtype=torch.float32)
        so = torch.empty(B, N, 2, device=dev, dtype=torch.float32) if want_stats else None
        N2 = 0 if W2 is None else W2.shape[0]
        y2 = None if W2 is None else torch.empty(B, M, N2, device=dev, dtype=torch.float32)
        self._check(self.L.imp_op_layer_gemm(self.handle, B, M, N, K, ks, _ptr(x), _ptr(x2), _ptr(W), _ptr(bias), _ptr(residual), _ptr(stats_in),
                                             _ptr(y), _ptr(so), _ptr(W2), _ptr(bias2), N2, _ptr(y2), int(pass_split), _stream(self.device)))
        return y, so, y2

    def op_fused_mlp(self, x, a, W0, b0, W3, b3, W2=None, b2=None, fake=False):
        """csrc/gemm_wf.hip's fused layer MLP on its own (include/imp_hip.h imp_op_fused_mlp): x, a [B, M, 256] -> y [B, M, 256]
        (+ the chained projection y2 [B, M, N2])"""
        x, a, W0, b0, W3, b3 = (_f32(t, n) for t, n in ((x, 'x'), (a, 'a'), (W0, 'W0'), (b0, 'b0'), (W3, 'W3'), (b3, 'b3')))
        B, M, _ = x.shape
        y = torch.empty(B, M, 256, device=x.device, dtype=torch.float32)
        N2 = 0 if W2 is None else W2.shape[0]
        W2 = None if W2 is None else _f32(W2, 'W2')
        b2 = None if b2 is None else _f32(b2, 'b2')
        y2 = None if W2 is None else torch.empty(B, M, N2, device=x.device, dtype=torch.float32)
        self._check(self.L.imp_op_fused_mlp(self.handle, B, M, _ptr(x), _ptr(a), _ptr(W0), _ptr(b0), _ptr(W3), _ptr(b3), _ptr(W2), _ptr(b2), N2,
                                            _ptr(y), _ptr(y2), int(bool(fake)), _stream(self.device)))
        return y, y2

    def op_attention(self, qkv_q, qkv_kv, key_mask=None, want_lse=True):
        qkv_q, qkv_kv = _f32(qkv_q, 'qkv_q'), _f32(qkv_kv, 'qkv_kv')
        B, nq, D3 = qkv_q.shape
        nk, D = qkv_kv.shape[1], D3 // 3
        out = torch.empty(B, nq, D, device=qkv_q.device, dtype=torch.float32)
        lse = torch.empty(B, 4, nq, device=qkv_q.device, dtype=torch.float32) if want_lse else None
        km = None if key_mask is None else key_mask.to(torch.uint8).contiguous()
        self._check(self.L.imp_op_attention(self.handle, B, nq, nk, D, _ptr(qkv_q), _ptr(qkv_kv), _ptr(km), _ptr(out),
                                            _ptr(lse), _stream(self.device)))
        return out, lse

    def time_attention(self, batch, n, reps):
        ms = C.c_float()
        self._check(self.L.imp_time_attention(self.handle, batch, n, reps, C.byref(ms), _stream(self.device)))
        return ms.value

    def time_attention_clock(self, batch, n, reps):
        """-> (ms per launch, shader clock in MHz the launches ran at; include/imp_hip.h imp_time_attention_clock)"""
        ms, mhz = C.c_float(), C.c_float()
        self._check(self.L.imp_time_attention_clock(self.handle, batch, n, reps, C.byref(ms), C.byref(mhz), _stream(self.device)))
        return ms.value, mhz.value

    def time_sinkhorn(self, batch, n, iterations):
        ms = C.c_float()
        self._check(self.L.imp_time_sinkhorn(self.handle, batch, n, iterations, C.byref(ms), _stream(self.device)))
        return ms.value

    def resident_status(self):
        """(timed_out, used) of the chip-resident Sinkhorn kernel on this context (synchronises)"""
        st, used = C.c_int(), C.c_int()
        self._check(self.L.imp_resident_status(self.handle, C.byref(st), C.byref(used)))
        return bool(st.value), bool(used.value)

    def resident_health(self, raise_on_timeout=True):
        """non-synchronising look at the health word of the chip-resident Sinkhorn kernel (call it after the results of a
        step were synchronised to the host): raises ResidentSinkhornTimeout when a launch since the last look was voided (or
        returns False), having already recovered the context; -> (voided launches so far, protocol level 0 / 1 / 2)"""
        n, lvl = C.c_int(), C.c_int()
        rc = self.L.imp_resident_health(self.handle, C.byref(n), C.byref(lvl))
        if rc == IMP_E_RESIDENT and not raise_on_timeout:
            return False
        self._check(rc)
        return n.value, lvl.value

    def option(self, name, value):
        """a named switch of the context, before its first compute call (include/imp_hip.h imp_ctx_option: step-down paths forced for A/B tests, fault hooks)"""
        self._check(self.L.imp_ctx_option(self.handle, name.encode(), C.c_long(int(value))))

    def resident_repaired(self):
        """calls whose own voided waiting launch (resident Sinkhorn, fused layer) was repaired inside the call (include/imp_hip.h imp_resident_repaired)"""
        return int(self.L.imp_resident_repaired(self.handle))

    POSTMORTEM_FIELDS = ('kind', 'launch_tag', 'block', 'hw_id', 'xcc', 'phase', 'waited_for', 'tag_expected', 'tag_seen', 'iteration', 'pair', 'group',
                         'groups', 'placement', 'pairs', None, 'status_word', 'gate_multi_stream', 'several_streams_choosing', 'gate_same_stream_run',
                         'level_before', 'fused_layers_on', 'next_sinkhorn_tag', 'last_fused_tag', 'voided_so_far')

    def resident_postmortem(self):
        """record of the LAST voided waiting launch of this context as a dict (include/imp_hip.h imp_resident_postmortem), or None"""
        buf = (C.c_int32 * 40)()
        if self.L.imp_resident_postmortem(self.handle, buf, 40) != 1:
            return None
        rec = {k: int(buf[i]) for i, k in enumerate(self.POSTMORTEM_FIELDS) if k}
        hw = rec['hw_id']
        rec['waiter'] = {'wave': hw & 15, 'simd': (hw >> 4) & 3, 'cu': (hw >> 8) & 15, 'sh': (hw >> 12) & 1, 'se': (hw >> 13) & 7}
        return rec

    def tag_wraps(self):
        """how often the tag counter of hipGraph-replayed resident launches wrapped (the library then cleared the exchange buffers)"""
        return int(self.L.imp_tag_wraps(self.handle))

    def set_resident_verify(self, on: bool):
        """every chip-resident Sinkhorn launch is awaited inside the call and a voided one is re-run there (one host
        synchronisation per score; calls then always return valid results)"""
        self._check(self.L.imp_set_resident_verify(self.handle, 1 if on else 0))

    def time_layer_gemm(self, batch, n, which, dbg=-1, reps=20):
        ms = C.c_float()
        self._check(self.L.imp_time_layer_gemm(self.handle, batch, n, which, dbg, reps, C.byref(ms), _stream(self.device)))
        return ms.value
