"""pose accuracy of the GPU pose step on synthetic two-view scenes - round 2's solver (eight-point sampler, inlier counting) against the
sigma-marginalised MAGSAC++ quality + IRLS refinement, on the eight-point and on the five-point sampler (round 3 default).  For every (outlier ratio, pixel noise) cell:
AUC@5 / @10 of max(err_R, err_t) over `scenes` scenes of 800 correspondences (tools/utils.py:445-457 definition), and ms per call.
OpenCV's USAC_MAGSAC itself is not available here: these numbers compare the two rankings, they do not pin the reference's solver."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imp_release_amd import metrics, pose as gpose
from oracle import pose_oracle as po
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
MODES = (('8pt', 'count', 4096), ('8pt', 'magsac', 4096), ('5pt', 'magsac', 1024), ('5pt', 'magsac', 256))
print('outliers noise | ' + ' | '.join('%s+%s x%d: AUC@5 AUC@10 found ms' % m for m in MODES))
for outl in (0.2, 0.4, 0.5, 0.6):
    for noise in (0.2, 0.5, 1.0):
        row = []
        for sampler, scoring, its in MODES:
            errs, t0 = [], time.perf_counter()
            for s in range(scenes):
                k0, k1, K, R, t, truth = po.synthetic_scene(800, outliers=outl, noise=noise, seed=1000 + s, angle_deg=6 + (s % 10) * 2)
                r = gpose.estimate_pose(k0, k1, K, K, 1.0, scoring=scoring, sampler=sampler, iterations=its)
                if r is None:
                    errs.append(np.inf); continue
                et, eR = metrics.compute_pose_error(np.hstack([R, t[:, None]]), r[1], r[2])
                errs.append(max(et, eR))
            ms = (time.perf_counter() - t0) / scenes * 1e3
            auc = metrics.pose_auc(errs, [5, 10])
            row.append('%5.1f %5.1f  %4.2f %5.2f' % (100 * auc[0], 100 * auc[1], np.mean(np.isfinite(errs)), ms))
        print('  %.1f    %.1f  |  ' % (outl, noise) + '  |  '.join(row))
