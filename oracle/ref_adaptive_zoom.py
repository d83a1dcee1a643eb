"""CPU oracle for the AdaptiveZoom row -- TEST INFRASTRUCTURE.  Only tests/ may import this; the product
(gyroflow_amd/adaptive_zoom.py) never does.

A frame-by-frame restatement of the reference's AdaptiveZoom, in the order the reference runs it:
    bounding_polygon   /root/reference/adaptive_zoom.py:193-228 (32 boundary points -> cv2.fisheye.undistortPoints :220,
                       here oracle/cv_oracle.c's undistort_points)
    find_fcorr         adaptive_zoom.py:33-62
    find_fov           adaptive_zoom.py:64-86 (scipy interp1d 'quadratic' through three points, as the reference)
    compute            adaptive_zoom.py:88-155 (rolling minimum + Gaussian smoothing of the per-frame FOV)
Pinned by the reference's own data: the fcorr column / min_fcorr of plugins/blender/GX016015.MP4.gyroflow
(tests/test_oracle.py::test_undistort_points_pinned_by_gyroflow_sample)."""
import numpy as np
from scipy.interpolate import interp1d
from scipy.spatial.transform import Rotation


def bounding_polygon(orc, K, D, calib_dimension, quat, numPoints=9):
    (W, H) = calib_dimension
    R = np.eye(3)
    if quat is not None:
        quat = np.asarray(quat, np.float64).flatten()
        R = Rotation.from_quat([quat[1], quat[2], quat[3], quat[0]]).as_matrix()
        R[[0, 0, 1, 2], [1, 2, 0, 0]] *= -1
    p = []
    for i in range(numPoints - 1):
        p.append((i * (W / (numPoints - 1)), 0))
    for i in range(numPoints - 1):
        p.append((W, i * (H / (numPoints - 1))))
    for i in range(numPoints - 1):
        p.append(((numPoints - 1 - i) * (W / (numPoints - 1)), H))
    for i in range(numPoints - 1):
        p.append((0, (numPoints - 1 - i) * (H / (numPoints - 1))))
    return orc.undistort_points(np.array(p, np.float64), K, D, R, K)


def find_fcorr(center, polygon, output_dim):
    (ow, oh) = output_dim
    angle_output = np.arctan2(oh / 2, ow / 2)
    polygon = polygon - center
    distP = np.linalg.norm(polygon, axis=1)
    angles = np.abs(np.arctan2(polygon[:, 1], polygon[:, 0]))
    mask = (angle_output <= np.abs(angles)) & (np.abs(angles) < (np.pi - angle_output))
    dWidth = np.abs((ow / 2) / np.cos(angles))
    dHeight = np.abs((oh / 2) / np.sin(angles))
    ffactor = dWidth / distP
    ffactor[mask] = dHeight[mask] / distP[mask]
    return np.max(ffactor), np.argmax(ffactor)


def find_fov(center, polygon, output_dim, numIntPoints=20):
    fcorr, idx = find_fcorr(center, polygon, output_dim)
    nP = polygon.shape[0]
    relevantP = polygon[((idx - 1) % nP, idx, (idx + 1) % nP), :]
    distance = np.cumsum(np.sqrt(np.sum(np.diff(relevantP, axis=0) ** 2, axis=1)))
    distance = np.insert(distance, 0, 0) / distance[-1]
    alpha = np.linspace(0, 1, numIntPoints)
    interpolated_points = interp1d(distance, relevantP, kind='quadratic', axis=0)(alpha)
    fcorrI, _ = find_fcorr(center, interpolated_points, output_dim)
    return 1 / np.max((fcorr, fcorrI))


def _gaussian(M, std):
    n = np.arange(0, M) - (M - 1.0) / 2.0
    return np.exp(-n ** 2 / (2 * std * std))


def compute(orc, K, D, calib_dimension, quaternions, output_dim, fps, smoothingFocus=2.0, tstart=False, tend=False):
    sff = int(smoothingFocus * fps)
    if sff % 2 == 0:
        sff += 1
    polys = [bounding_polygon(orc, K, D, calib_dimension, q) for q in quaternions]
    centers = np.array([(calib_dimension[0] / 2, calib_dimension[1] / 2) for q in quaternions])
    fov = np.array([find_fov(c, p, output_dim) for c, p in zip(centers, polys)])
    if tend != False:  # noqa: E712
        max_fov = np.max(fov)
        fov[:max(tstart, 0)] = max_fov
        fov[tend:] = max_fov
    if smoothingFocus > 0:
        g = _gaussian(sff, sff / 6)
        g = g / np.sum(g)
        pad = np.pad(fov, (int(sff / 2), int(sff / 2)), mode='edge')
        fmin = np.min(np.lib.stride_tricks.sliding_window_view(pad, sff), axis=1)
        fov = np.convolve(np.pad(fmin, (int(sff / 2), int(sff / 2)), mode='edge'), g, 'valid')
    elif smoothingFocus == -1:
        fov = np.repeat(np.min(fov), fov.size)
    elif smoothingFocus == -2:
        fov = np.repeat(1, fov.size)
    return fov, centers, polys
