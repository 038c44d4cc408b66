"""SuperPoint front-end on the GPU: host mirror of ``nets/superpoint.py``.

``SuperPoint(config)`` takes the reference's config keys (nets/superpoint.py:104-110 + ``weight_path``) and its ``forward(data)``
/ ``extract(data)`` return what the reference returns: per-image lists ``keypoints`` [N, 2] (x, y) float32, ``scores`` [N],
``descriptors`` [D, N] - CUDA tensors produced by csrc/superpoint.hip through the C-ABI (``imp_sp_*`` in include/imp_hip.h).
No CPU path: without the HIP library construction raises ``HipLibraryMissing``.

Weights: ``config['weight_path']`` (a ``torch.save``d state_dict such as superpoint_v1.pth, nets/superpoint.py:155-156) or
``config['state_dict']`` (mapping name -> tensor / ndarray; the seeded random weights of ``synthetic.make_superpoint_state_dict``
in the tests - the released checkpoint is not available offline).

``align_corners`` of ``sample_descriptors``: the reference passes ``align_corners=True`` to ``grid_sample`` only when
``int(torch.__version__[2]) > 2`` (nets/superpoint.py:89).  Character 2 of the version string is the first digit of the MINOR
version, so the test is True for torch 1.3 ... 1.9 and 2.3 ... 2.9 and False for x.0 ... x.2 and for every two-digit minor
(1.10 ... 1.13, 2.10 ... - including the 2.10 of this image: ``'2.10.0'[2] == '1'``); False means ``grid_sample``'s default
``align_corners=False``.  ``config['align_corners']`` = None (default) applies the same rule to the installed torch, so both
implementations agree wherever they are run side by side; True / False force either behaviour.  Both branches are pinned by
reference-generated fixtures (``superpoint_aligned_*`` captured with the version string patched), and
``tests/test_host_cpu.py::test_superpoint_align_corners_rule`` pins the rule itself on version strings.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def reference_align_corners(version: str = None) -> bool:
    """the rule of nets/superpoint.py:89 on a torch version string"""
    v = torch.__version__ if version is None else version
    try:
        return int(str(v)[2]) > 2
    except (ValueError, IndexError):
        return False


class SuperPoint(torch.nn.Module):
    default_config = {
        'descriptor_dim': 256,
        'nms_radius': 4,
        'keypoint_threshold': 0.0025,
        'max_keypoints': -1,
        'remove_borders': 4,
        'align_corners': None,
    }

    def __init__(self, config, device=None):
        super().__init__()
        self.config = {**self.default_config, **config}
        mk = self.config['max_keypoints']
        if mk == 0 or mk < -1:
            raise ValueError('"max_keypoints" must be positive or "-1"')          # nets/superpoint.py:161-163
        if 'state_dict' in self.config and self.config['state_dict'] is not None:
            sd = self.config['state_dict']
        else:
            sd = torch.load(str(self.config['weight_path']), map_location='cpu')
        self._L = _lib.lib()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('SuperPoint runs on the GPU only (libimp_hip has no CPU path)')
        h = C.c_void_p()
        self._check(self._L.imp_sp_create(C.byref(h), self.device.index or 0, int(self.config['descriptor_dim'])))
        self._h = h
        for name, t in sd.items():
            a = np.ascontiguousarray((t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).astype(np.float32))
            self._check(self._L.imp_sp_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        self._check(self._L.imp_sp_finalize(self._h))

    def _check(self, rc):
        if rc != 0:
            raise _lib.ImpError(rc, self._L.imp_last_error().decode())

    def __del__(self):
        try:
            h = self.__dict__.get('_h')
            if h is not None:
                self.__dict__['_h'] = None
                self.__dict__['_L'].imp_sp_destroy(h)
        except Exception:           # noqa: BLE001  (interpreter shutdown)
            pass

    def _detect(self, data):
        image = data['image']
        if image.dim() != 4 or image.shape[1] != 1:
            raise ValueError('image must be [B, 1, H, W]')
        img = _lib._f32(image.to(self.device), 'image')
        B, _, H, W = img.shape
        counts = (C.c_int * B)()
        ac = self.config['align_corners']
        ac = reference_align_corners() if ac is None else bool(ac)
        st = _lib._stream(self.device)
        self._check(self._L.imp_sp_detect(self._h, _lib._ptr(img), B, H, W, int(self.config['nms_radius']),
                                          float(self.config['keypoint_threshold']), int(self.config['max_keypoints']),
                                          int(self.config['remove_borders']), int(ac), st, counts))
        return img, list(counts), st

    @torch.no_grad()
    def forward(self, data):
        """nets/superpoint.py:170-232"""
        img, counts, st = self._detect(data)
        D = int(self.config['descriptor_dim'])
        out = {'keypoints': [], 'scores': [], 'descriptors': []}
        for b, n in enumerate(counts):
            kp = torch.empty((n, 2), dtype=torch.float32, device=self.device)
            sc = torch.empty((n,), dtype=torch.float32, device=self.device)
            de = torch.empty((D, n), dtype=torch.float32, device=self.device)
            self._check(self._L.imp_sp_describe(self._h, b, _lib._ptr(kp), _lib._ptr(sc), _lib._ptr(de), st))
            out['keypoints'].append(kp)
            out['scores'].append(sc)
            out['descriptors'].append(de)
        return out

    @torch.no_grad()
    def extract(self, data, nms: bool = False):
        """nets/superpoint.py:140-168 -> (scores [B, 8h, 8w], descriptors [B, D, h, w]); nms=True returns the map after
        simple_nms instead (a probe the reference does not have)"""
        img, _, st = self._detect(data)
        B, _, H, W = img.shape
        h, w = H // 2 // 2 // 2, W // 2 // 2 // 2
        D = int(self.config['descriptor_dim'])
        scores = torch.empty((B, h * 8, w * 8), dtype=torch.float32, device=self.device)
        desc = torch.empty((B, D, h, w), dtype=torch.float32, device=self.device)
        self._check(self._L.imp_sp_dense(self._h, None if nms else _lib._ptr(scores), _lib._ptr(scores) if nms else None,
                                         _lib._ptr(desc), st))
        return scores, desc

    LAYERS = ['conv1a', 'conv1b', 'conv2a', 'conv2b', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'convPa|convDa', 'convDb']

    @torch.no_grad()
    def op_conv(self, layer: int, x, relu: bool = True, pool: bool = False):
        """test entry (imp_sp_op_conv): one convolution of the stack; x NCHW float32 -> NCHW (NHWC inside, as the kernels keep it)"""
        x = _lib._f32(x.to(self.device), 'x')
        B, Cin, H, W = x.shape
        xin = x[:, 0].contiguous() if layer == 0 else x.permute(0, 2, 3, 1).contiguous()
        cout = [64, 64, 64, 64, 128, 128, 128, 128, 512, int(self.config['descriptor_dim'])][layer]
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=self.device)
        self._check(self._L.imp_sp_op_conv(self._h, layer, _lib._ptr(xin), B, H, W, _lib._ptr(out), int(relu), int(pool),
                                           _lib._stream(self.device)))
        return out.permute(0, 3, 1, 2).contiguous()
