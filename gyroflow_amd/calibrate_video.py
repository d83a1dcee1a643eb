"""FisheyeCalibrator: lens model holder + per-frame map generation, on MI355X.

Host-side mirror of the hot-path part of the reference class of the same name
(/root/reference/calibrate_video.py:33): preset loading (:520-587, :642-652), get_maps (:330-384) and
undistort_image (:296-328).
The chessboard calibration itself (cv2.fisheye.calibrate, :137-259) is out of scope.

get_maps() keeps the reference signature and returns the same (map1, map2) pair -- as CUDA tensors -- by calling
libgfremap's map kernels; frame_xform()/frame_xforms() expose the same per-frame parameters for the fused
kernel that never materialises the maps.
"""
import json

import numpy as np
from scipy.spatial.transform import Rotation

from . import _lib, cv

__version__ = "0.3.0-beta"      # preset compatibility tag of the reference (_version.py)


def _quat_to_R(quat):
    """(w,x,y,z) -> 3x3 with the reference's axis flips (calibrate_video.py:351-356)."""
    q = np.asarray(quat, dtype=np.float64).flatten()
    R = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
    R[[0, 0, 1, 2], [1, 2, 0, 0]] *= -1
    return R


class FisheyeCalibrator:
    def __init__(self, chessboard_size=(9, 6)):
        self.chessboard_size = chessboard_size
        self.orig_dimension = np.array([0, 0])     # equal to calib_dimension when the footage is not stretched
        self.calib_dimension = np.array([0, 0])
        self.K = np.zeros((3, 3))
        self.D = np.zeros((4, 1))
        self.RMS_error = 100
        self.input_horizontal_stretch = 1
        self.data_from_preset_file = False
        self.extra_cam_info = None
        self.num_images = 0
        self.num_images_used = 0
        self.new_K = None

    # ------------------------------------------------------------------ stretch helpers (:96-116)
    def set_horizontal_stretch(self, new_stretch=1):
        self.input_horizontal_stretch = new_stretch

    def image_is_stretched(self):
        return self.input_horizontal_stretch != 1

    def get_stretched_size_from_dimension(self, new_img_dim):
        return (round(new_img_dim[0] * self.calib_dimension[0] / self.orig_dimension[0]),
                round(new_img_dim[1] * self.calib_dimension[1] / self.orig_dimension[1]))

    # ------------------------------------------------------------------ presets (:520-587, :642-652)
    # Descriptive preset fields copied into extra_cam_info: (key, default); None = the field is taken as it is (may be absent).
    _INFO_FIELDS = (("note", None), ("calibrated_by", "N/A"), ("camera_brand", "N/A"), ("camera_model", "N/A"),
                    ("camera_setting", "N/A"), ("lens_model", "N/A"), ("calibrator_version", None), ("date", None))

    def load_calibration_data(self, cal_data, printinfo=False):
        """Take K, D and the calibration / original frame sizes from a preset dictionary (the JSON schema of
        calibrate_video.py:483-518 as read at :520-587); returns the descriptive fields as extra_cam_info.

        Same order of events as the reference: the lens-model check, the optional summary and the version note (printed whether
        or not printinfo is set, :536-538; a preset without "calibrator_version" raises KeyError there) all come before
        anything but data_from_preset_file changes, so a preset that fails them leaves K and D as they were."""
        if not cal_data["use_opencv_fisheye"]:
            raise Exception("Preset not for OpenCV fisheye lens model")
        self.data_from_preset_file = True
        self._announce_preset(cal_data, printinfo)

        wh = lambda d: (d["w"], d["h"])
        lens = cal_data["fisheye_params"]
        self.calib_dimension = wh(cal_data["calib_dimension"])
        stretched = "orig_dimension" in cal_data                         # preset format >= 0.3.0
        if stretched:
            orig = wh(cal_data["orig_dimension"])
            self.input_horizontal_stretch = cal_data["input_horizontal_stretch"]
            self.orig_dimension = orig
        else:
            self.input_horizontal_stretch = 1
            self.orig_dimension = self.calib_dimension
        self.num_images = self.num_images_used = cal_data["num_images"]
        self.RMS_error = lens["RMS_error"]
        self.K = np.array(lens["camera_matrix"])
        self.D = np.array(lens["distortion_coeffs"])

        info = {"name": " ".join(cal_data.get("name").replace("_", " ").split())}
        for key, default in self._INFO_FIELDS:
            info[key] = cal_data.get(key) if default is None else cal_data.get(key, default)
        ow, oh = self.orig_dimension
        if oh == 0:
            raise KeyError("Error loading preset file")
        info.update(width=ow, height=oh, aspect=ow / oh, num_images=self.num_images)
        self.extra_cam_info = info
        return info

    @staticmethod
    def _announce_preset(cal_data, printinfo):
        """The console lines of calibrate_video.py:528-538."""
        if printinfo:
            print("Preset name: {}".format(cal_data["name"]))
            print("Note: {}".format(cal_data["note"]))
            print("Made with {} frames using calibrator version {} on date {}".format(
                cal_data["num_images"], cal_data["calibrator_version"], cal_data["date"]))
        preset_version = cal_data["calibrator_version"]
        if preset_version != __version__:
            print("Note: Versions don't match. Calibrator: {}, Preset: {}. Should be fine though.".format(__version__, preset_version))

    def get_calibration_data(self):
        """Preset dictionary in the reference's format (calibrate_video.py:483-518)."""
        info = getattr(self, "extra_cam_info", {}) or {}
        return {
            "name": info.get("name", ""), "note": info.get("note", ""), "calibrated_by": info.get("calibrated_by", "N/A"),
            "camera_brand": info.get("camera_brand", "N/A"), "camera_model": info.get("camera_model", "N/A"),
            "lens_model": info.get("lens_model", "N/A"), "camera_setting": info.get("camera_setting", "N/A"),
            "calibrator_version": info.get("calibrator_version", "N/A"), "date": info.get("date", "N/A"),
            "calib_dimension": {"w": self.calib_dimension[0], "h": self.calib_dimension[1]},
            "orig_dimension": {"w": self.orig_dimension[0], "h": self.orig_dimension[1]},
            "input_horizontal_stretch": self.input_horizontal_stretch,
            "num_images": self.num_images_used,
            "use_opencv_fisheye": True,
            "fisheye_params": {"RMS_error": self.RMS_error, "camera_matrix": np.asarray(self.K).tolist(),
                               "distortion_coeffs": np.asarray(self.D).flatten().tolist()},
            "use_opencv_standard": False,
            "calib_params": {},
        }

    def load_calibration_json(self, filename, printinfo=False):
        with open(filename, "r") as infile:
            presets = json.load(infile)
        return self.load_calibration_data(presets, printinfo)

    # ------------------------------------------------------------------ one-shot undistortion (:296-328)
    def undistort_image(self, img, fov_scale=1.0, device=None):
        """Undistort one image with the fisheye model (reference: calibrate_video.py:296-328): new camera matrix from
        cv2.fisheye.estimateNewCameraMatrixForUndistortRectify (:314, sets self.new_K), CV_16SC2 maps with R = I (:322),
        bilinear remap with a constant black border (:325).  The reference first re-runs the chessboard calibration
        (compute_calibration, out of scope here): K and D are taken as loaded.  numpy in -> numpy out, tensor in -> tensor out."""
        img_dim = (int(img.shape[1]), int(img.shape[0]))
        scaled_K = self.K * img_dim[0] / self.calib_dimension[0]
        scaled_K[2][2] = 1.0
        new_K = cv.fisheye_estimateNewCameraMatrixForUndistortRectify(scaled_K, self.D, img_dim, np.eye(3), fov_scale=fov_scale,
                                                                      device=device)
        self.new_K = new_K
        map1, map2 = cv.fisheye_initUndistortRectifyMap(scaled_K, self.D, np.eye(3), new_K, img_dim, cv.CV_16SC2, device=device)
        return cv.remap(img, map1, map2, cv.INTER_LINEAR, borderMode=cv.BORDER_CONSTANT)

    # ------------------------------------------------------------------ per-frame parameters (:341-370)
    def _frame_matrices(self, fov_scale, output_dim, new_img_dim, quat, focalCenter, original_stretched):
        """The (scaled_K, D, R, new_K, out_dim) the reference hands to cv2.fisheye.initUndistortRectifyMap."""
        if new_img_dim and self.image_is_stretched() and original_stretched:
            # new_img_dim is the size of the unstretched video frame
            new_img_dim = (round(new_img_dim[0] * self.calib_dimension[0] / self.orig_dimension[0]),
                           round(new_img_dim[1] * self.calib_dimension[1] / self.orig_dimension[1]))
        img_dim = new_img_dim if new_img_dim else self.calib_dimension
        out_dim = output_dim if output_dim else self.calib_dimension
        if focalCenter is None:
            focalCenter = np.array([self.calib_dimension[0] / 2, self.calib_dimension[1] / 2])

        R = np.eye(3) if quat is None else _quat_to_R(quat)

        img_dim_ratio = img_dim[0] / self.calib_dimension[0]
        scaled_K = self.K * img_dim_ratio
        scaled_K[2][2] = 1.0

        new_K = np.copy(self.K)
        new_K[0][0] = new_K[0][0] * 1.0 / fov_scale
        new_K[1][1] = new_K[1][1] * 1.0 / fov_scale
        new_K[0][2] = (self.calib_dimension[0] / 2 - focalCenter[0]) * img_dim_ratio / fov_scale + out_dim[0] / 2
        new_K[1][2] = (self.calib_dimension[1] / 2 - focalCenter[1]) * img_dim_ratio / fov_scale + out_dim[1] / 2
        return scaled_K, self.D, R, new_K, (int(out_dim[0]), int(out_dim[1]))

    def get_maps(self, fov_scale=1.0, output_dim=None, new_img_dim=None, update_new_K=True, quat=None,
                 focalCenter=None, original_stretched=True, device=None):
        """Undistortion + rotation maps for one frame (reference: calibrate_video.py:330).

        Returns (map1 int16[H,W,2], map2 uint16[H,W]) CUDA tensors in cv2's CV_16SC2 / CV_16UC1 fixed-point layout."""
        scaled_K, D, R, new_K, out_dim = self._frame_matrices(fov_scale, output_dim, new_img_dim, quat, focalCenter,
                                                             original_stretched)
        if update_new_K:
            self.new_K = new_K
        if original_stretched and self.image_is_stretched():
            fx, fy = cv.fisheye_initUndistortRectifyMap(scaled_K, D, R, new_K, out_dim, cv.CV_32FC1, device=device)
            # rescale to the stretched video's pixel grid, then to fixed point (float32, as numpy does it at :378)
            return cv.convertMaps(fx, fy, cv.CV_16SC2,
                                  scale_x=(self.orig_dimension[0], self.calib_dimension[0]),
                                  scale_y=(self.orig_dimension[1], self.calib_dimension[1]))
        return cv.fisheye_initUndistortRectifyMap(scaled_K, D, R, new_K, out_dim, cv.CV_16SC2, device=device)

    # ------------------------------------------------------------------ fused-path parameters
    def map_scale(self):
        """The float32 map rescale of a stretched preset, (orig_w, calib_w, orig_h, calib_h) (calibrate_video.py:378), for
        cv.undistort_rotate(map_scale=...); None for ordinary presets."""
        if not self.image_is_stretched():
            return None
        return (self.orig_dimension[0], self.calib_dimension[0], self.orig_dimension[1], self.calib_dimension[1])

    def frame_xform(self, fov_scale=1.0, output_dim=None, new_img_dim=None, quat=None, focalCenter=None):
        """One gf_frame_xform row (13 float64) for cv.undistort_rotate (which also takes self.D): same parameters as get_maps().
        For a stretched preset pass map_scale() to cv.undistort_rotate as well."""
        scaled_K, D, R, new_K, _ = self._frame_matrices(fov_scale, output_dim, new_img_dim, quat, focalCenter, True)
        return _lib.xform_to_array(_lib.make_frame_xform(scaled_K, R, new_K))

    def frame_xforms(self, fov_scales, quats, output_dim=None, new_img_dim=None, focalCenters=None):
        """[N,13] float64: frame_xform() for every (fov_scale[i], quat[i], focalCenter[i]), built in one pass: the same float64
        operations as _frame_matrices (calibrate_video.py:351-370), element-wise over the batch, then one C call."""
        quats = np.asarray(quats, np.float64).reshape(-1, 4)
        n = quats.shape[0]
        fov = np.broadcast_to(np.asarray(fov_scales, np.float64), (n,))
        if new_img_dim and self.image_is_stretched():          # new_img_dim is the size of the unstretched video frame (:341-343)
            new_img_dim = self.get_stretched_size_from_dimension(new_img_dim)
        img_dim = new_img_dim if new_img_dim else self.calib_dimension
        out_dim = output_dim if output_dim else self.calib_dimension
        cd = self.calib_dimension
        if focalCenters is None:
            fc = np.broadcast_to(np.array([cd[0] / 2, cd[1] / 2]), (n, 2))
        else:
            fc = np.asarray(focalCenters, np.float64).reshape(n, 2)
        R = Rotation.from_quat(quats[:, [1, 2, 3, 0]]).as_matrix()
        R[:, [0, 0, 1, 2], [1, 2, 0, 0]] *= -1
        img_dim_ratio = img_dim[0] / cd[0]
        scaled_K = self.K * img_dim_ratio
        scaled_K[2][2] = 1.0
        P = np.tile(np.asarray(self.K, np.float64), (n, 1, 1))
        P[:, 0, 0] = P[:, 0, 0] * 1.0 / fov
        P[:, 1, 1] = P[:, 1, 1] * 1.0 / fov
        P[:, 0, 2] = (cd[0] / 2 - fc[:, 0]) * img_dim_ratio / fov + out_dim[0] / 2
        P[:, 1, 2] = (cd[1] / 2 - fc[:, 1]) * img_dim_ratio / fov + out_dim[1] / 2
        return _lib.make_frame_xforms(scaled_K, R, P)
