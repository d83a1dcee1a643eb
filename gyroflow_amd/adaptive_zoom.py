"""Mirror of the reference's AdaptiveZoom (/root/reference/adaptive_zoom.py:20-228): the dynamic per-frame FOV that
feeds `fcorr[n]` / `focalCenter[n]` to every get_maps call of the render loop (stabilizer.py:1338-1342, 1423-1427).

Same class, method names, arguments and return values.  What differs is the execution plan:
  * boundingPolygon for ALL frames is one launch of the undistortPoints kernel (F rotations x 32 points) instead of
    F calls of cv2.fisheye.undistortPoints;
  * findFcorr / findFov are evaluated for all frames at once with numpy array arithmetic (same formulas, same
    order of operations per element); the 3-point 'quadratic' interp1d of adaptive_zoom.py:75 is the unique
    parabola through the three points, evaluated in Lagrange form.
The per-frame methods (findFcorr, findFov, boundingPolygon, min_rolling) are kept with the reference's signatures;
findFocalCenter (adaptive_zoom.py:158-190) is dead code in the reference -- nothing calls it -- and has no mirror."""
import numpy as np
from scipy.spatial.transform import Rotation

from . import cv


def _gaussian(M, std):
    # scipy.signal.gaussian (adaptive_zoom.py:137; removed from newer scipy, same formula as signal.windows.gaussian)
    n = np.arange(0, M) - (M - 1.0) / 2.0
    return np.exp(-n ** 2 / (2 * std * std))


class AdaptiveZoom:
    def __init__(self, fisheyeCalibrator, device=None):
        self.calibrator = fisheyeCalibrator
        self.calib_dimension = fisheyeCalibrator.calib_dimension
        self.K = np.copy(fisheyeCalibrator.K)
        self.D = np.copy(fisheyeCalibrator.D)
        self.device = device

    # ---- adaptive_zoom.py:27-31: minimum over every window of `window` consecutive samples (last axis)
    def min_rolling(self, a, window, axis=1):
        return np.lib.stride_tricks.sliding_window_view(np.asarray(a), int(window), axis=-1).min(axis=axis)

    # ---- adaptive_zoom.py:33-62, for [..., N, 2] polygons and [..., 2] centres at once
    @staticmethod
    def _ffactor(center, polygon, output_dim):
        (output_width, output_height) = output_dim
        angle_output = np.arctan2(output_height / 2, output_width / 2)
        polygon = polygon - np.asarray(center)[..., None, :]
        distP = np.sqrt(polygon[..., 0] ** 2 + polygon[..., 1] ** 2)
        angles = np.abs(np.arctan2(polygon[..., 1], polygon[..., 0]))
        mask = (angle_output <= np.abs(angles)) & (np.abs(angles) < (np.pi - angle_output))
        with np.errstate(divide="ignore", invalid="ignore"):
            dWidth = np.abs((output_width / 2) / np.cos(angles))
            dHeight = np.abs((output_height / 2) / np.sin(angles))
            ffactor = np.where(mask, dHeight / distP, dWidth / distP)
        return ffactor

    def findFcorr(self, center, polygon, output_dim):
        ffactor = self._ffactor(np.asarray(center, np.float64), np.asarray(polygon, np.float64), output_dim)
        return np.max(ffactor), np.argmax(ffactor)

    # ---- adaptive_zoom.py:64-86
    def findFov(self, center, polygon, output_dim, numIntPoints=20):
        return float(self._find_fov_batch(np.asarray(center, np.float64)[None], np.asarray(polygon, np.float64)[None],
                                          output_dim, numIntPoints)[0])

    def _find_fov_batch(self, centers, polygons, output_dim, numIntPoints=20):
        F, nP, _ = polygons.shape
        ff = self._ffactor(centers, polygons, output_dim)                    # [F, nP]
        idx = np.argmax(ff, axis=1)
        fcorr = ff[np.arange(F), idx]
        rel = np.stack([polygons[np.arange(F), (idx - 1) % nP], polygons[np.arange(F), idx],
                        polygons[np.arange(F), (idx + 1) % nP]], axis=1)     # [F, 3, 2]
        seg = np.sqrt(np.sum(np.diff(rel, axis=1) ** 2, axis=2))            # [F, 2]
        total = seg[:, 0] + seg[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            x1 = seg[:, 0] / total                                           # knots 0, x1, 1
            alpha = np.linspace(0, 1, numIntPoints)[None, :]                 # [1, nI]
            x1c = x1[:, None]
            l0 = (alpha - x1c) * (alpha - 1.0) / (x1c * 1.0)                 # (a-x1)(a-1)/((0-x1)(0-1))
            l1 = alpha * (alpha - 1.0) / (x1c * (x1c - 1.0))
            l2 = alpha * (alpha - x1c) / (1.0 - x1c)
        interp = (l0[..., None] * rel[:, None, 0, :] + l1[..., None] * rel[:, None, 1, :] + l2[..., None] * rel[:, None, 2, :])
        ffI = self._ffactor(centers, interp, output_dim)
        fcorrI = np.max(ffI, axis=1)
        return 1 / np.maximum(fcorr, fcorrI)

    # ---- adaptive_zoom.py:88-155
    def compute(self, quaternions, output_dim, fps, smoothingFocus=2.0, tstart=False, tend=False, debug_plots=False,
                plot_blocking=False):
        # smoothingFocus == -1: totally disable; == -2: find minimum sufficient crop (reference comments :90-91)
        smoothingFocusFrames = int(smoothingFocus * fps)
        if smoothingFocusFrames % 2 == 0:
            smoothingFocusFrames = smoothingFocusFrames + 1

        quaternions = np.asarray(quaternions, np.float64).reshape(-1, 4)
        boundaryPolygons = self.boundingPolygons(quaternions)
        cropCenterPositions = np.array([(self.calib_dimension[0] / 2, self.calib_dimension[1] / 2) for q in quaternions])
        fovValues = self._find_fov_batch(cropCenterPositions, boundaryPolygons, output_dim)

        if tend != False:  # noqa: E712  (the reference's own test: tend=0 also means "not given")
            max_fov = np.max(fovValues)
            fovValues[:max(tstart, 0)] = max_fov
            fovValues[tend:] = max_fov

        if smoothingFocus > 0:
            filterCoeffFocus = _gaussian(smoothingFocusFrames, smoothingFocusFrames / 6)
            filterCoeffFocus = filterCoeffFocus / np.sum(filterCoeffFocus)
            half = int(smoothingFocusFrames / 2)
            fovValuesPad = np.pad(fovValues, (half, half), mode='edge')
            fovMin = self.min_rolling(fovValuesPad, window=smoothingFocusFrames)
            fovSmooth = np.convolve(np.pad(fovMin, (half, half), mode='edge'), filterCoeffFocus, 'valid')
            fovValues = fovSmooth
        elif smoothingFocus == -1:
            maxF = np.min(fovValues)
            fovValues = np.repeat(maxF, fovValues.size)
        elif smoothingFocus == -2:
            fovValues = np.repeat(1, fovValues.size)

        return fovValues, cropCenterPositions

    # ---- adaptive_zoom.py:193-216: numPoints - 1 equally spaced points per image edge, clockwise from the top-left corner
    def _distorted_points(self, numPoints=9):
        (w, h) = self.calib_dimension
        n = numPoints - 1
        fwd, back = np.arange(n, dtype=np.float64), np.arange(n, 0, -1, dtype=np.float64)
        xs = np.concatenate([fwd * (w / n), np.full(n, float(w)), back * (w / n), np.zeros(n)])
        ys = np.concatenate([np.zeros(n), fwd * (h / n), np.full(n, float(h)), back * (h / n)])
        return np.stack([xs, ys], axis=1)

    @staticmethod
    def _rotations(quats):
        quats = np.asarray(quats, np.float64).reshape(-1, 4)
        R = Rotation.from_quat(quats[:, [1, 2, 3, 0]]).as_matrix()       # Rotation([x,y,z,w]) normalises
        R[:, [0, 0, 1, 2], [1, 2, 0, 0]] *= -1
        return R

    def boundingPolygons(self, quaternions, numPoints=9):
        """All frames at once: [F,4] (w,x,y,z) -> [F, 4*(numPoints-1), 2]."""
        R = self._rotations(quaternions)
        return cv.fisheye_undistortPoints(self._distorted_points(numPoints), self.K, self.D, R=R, P=self.K, device=self.device)

    def boundingPolygon(self, quat, numPoints=9):
        if quat is None:
            return cv.fisheye_undistortPoints(self._distorted_points(numPoints), self.K, self.D, R=np.eye(3), P=self.K,
                                              device=self.device)
        return self.boundingPolygons(np.asarray(quat, np.float64).flatten()[None], numPoints)[0]
