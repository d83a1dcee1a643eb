"""Stabilizer: the render/export loop of the hot path, re-hosted on MI355X.

Mirror of the render-path part of /root/reference/stabilizer.py: `set_hyperlapse` (:239-248), `map_function`
(:1205-1215), `set_map_func_scale` (:1217) and `renderfile` (:1220-1531).  Video decode (cv2.VideoCapture) and
encode (vidgear/ffmpeg) stay on the CPU and are abstracted as a frame source / sink so synthetic frames can drive the
same loop; gyro parsing, integration, smoothing and optical-flow sync are out of scope -- the per-frame stabilisation
quaternions `stab_transform` (w,x,y,z per video frame) are an input.

Loop rules kept from the reference: the loop runs i = 1 .. num_frames-1 with
num_frames = int((stop-start)*fps); frame_num is the source position *before* the read; a group of
`hyperlapse_multiplier` frames contributes its first `hyperlapse_num_blended_frames` frames; the output frame is
emitted on the group's last blended frame; rendering stops when frame_num runs past stab_transform or after more
than five consecutive failed reads.
"""
import numpy as np
import torch
from matplotlib import colors

from . import cv
from .adaptive_zoom import AdaptiveZoom
from .calibrate_video import FisheyeCalibrator
from .pipeline import BatchPipeline


# ----------------------------------------------------------------------------- frame sources / sinks
class ArraySource:
    """Frame source over an in-memory sequence of HxWx3 arrays (stands in for cv2.VideoCapture).
    `ring=True` cycles the sequence so a few distinct frames can represent a long clip."""

    def __init__(self, frames, count=None, ring=False, fail_at=()):
        self.frames = frames
        self.count = len(frames) if count is None else int(count)
        self.ring = ring
        self.pos = 0
        self.fail_at = set(fail_at)

    def get_pos(self):
        return self.pos

    def set_pos(self, n):
        self.pos = int(n)

    def read(self):
        p = self.pos
        if p >= self.count:
            return False, None
        self.pos += 1
        if p in self.fail_at:
            return False, None
        return True, self.frames[p % len(self.frames) if self.ring else p]


def synthetic_frames(width, height, n=8, dtype=np.uint8, seed=1234):
    """Seeded synthetic BGR frames: smooth gradient + 16-px checkerboard + +-8 noise.
    uint16 frames carry 10-bit values shifted left by 6."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    frames = []
    for k in range(n):
        base = np.stack([(xx * 255.0 / max(width - 1, 1) + 17 * k) % 256,
                         (yy * 255.0 / max(height - 1, 1) + 29 * k) % 256,
                         ((xx + yy) * 255.0 / max(width + height - 2, 1) + 43 * k) % 256], axis=-1)
        checker = ((((xx // 16) + (yy // 16) + k) & 1) * 48.0)[..., None]
        noise = rng.integers(-8, 9, size=(height, width, 3))
        img = np.clip(base * 0.75 + checker + noise, 0, 255)
        if np.dtype(dtype) == np.uint16:
            frames.append((np.clip(img * 4.0 + rng.integers(0, 4, size=img.shape), 0, 1023).astype(np.uint16) << 6))
        else:
            frames.append(img.astype(np.uint8))
    return frames


class ListSink:
    def __init__(self):
        self.frames = []

    def write(self, frame):
        self.frames.append(np.array(frame, copy=True))

    def close(self):
        pass


class NullSink:
    """Counts frames and keeps a cheap checksum (sum of all bytes mod 2^64) instead of storing them."""

    def __init__(self):
        self.count = 0
        self.checksum = 0

    def write(self, frame):
        self.count += 1
        self.checksum = (self.checksum + int(frame[::16, ::16].astype(np.uint64).sum())) & 0xFFFFFFFFFFFFFFFF

    def close(self):
        pass


def frame_shard(n_frames, rank, world, multiple=1):
    """Contiguous frame range [lo, hi) of `rank` when a clip of n_frames is split over `world` GPUs.

    Frames are independent given their own (quat, fcorr, focalCenter) (stabilizer.py:1423-1443), so ranks need no
    exchange step; shard boundaries are aligned to `multiple` (= hyperlapse_multiplier) so that no blend group
    (stabilizer.py:1431-1468) straddles two ranks.  bg_color="HISTORY" makes frame n depend on output n-1 and does not
    shard."""
    units = -(-n_frames // multiple)
    lo = (units * rank // world) * multiple
    hi = min((units * (rank + 1) // world) * multiple, n_frames)
    return lo, hi


# ----------------------------------------------------------------------------- the stabilizer
class Stabilizer:
    def __init__(self, source, calibration, stab_transform, fps, width, height, device=None, video_rotation=-1):
        """source: frame source; calibration: FisheyeCalibrator or preset path; stab_transform: [F,4] (w,x,y,z);
        width/height: size of the video frames (`orig_dimension` in the reference); video_rotation: a cv2.rotate code applied
        to every decoded frame before the remap (reference :89,153-164,1385-1386), -1 for none.  As in the reference (:164-165
        swaps nothing), orig_dimension stays the decoded size for the 90 degree codes too."""
        self.cap = source
        self.video_rotate_code = video_rotation
        if isinstance(calibration, FisheyeCalibrator):
            self.undistort = calibration
        else:
            self.undistort = FisheyeCalibrator()
            self.undistort.load_calibration_json(calibration)
        self.stab_transform = None if stab_transform is None else np.asarray(stab_transform, np.float64)
        self.times = None
        self.new_integrator = self.integrator = None
        self.smoothing_algo = None
        self.gyro_data = self.acc_data = None
        self.fps = fps
        self.width, self.height = int(width), int(height)
        self.orig_dimension = (self.width, self.height)
        self.map_func_scale = 0.9
        self.hyperlapse_multiplier = 1
        self.hyperlapse_num_blended_frames = 1
        self.hyperlapse_skipped_frames = 0
        if device is not None:
            self.device = torch.device(device)
        else:       # no GPU: only the host-side logic (schedule(), border selection) is usable
            self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.adaptive_zoom = None          # optional override: (stab_transform, out_size, fps, smoothingFocus, tstart, tend) -> (fcorr, focalCenter)

    # ------------------------------------------------------------------ quaternion producer + .gyroflow data files
    def set_smoothing_algo(self, algo=None):
        from .smoothing_algos import PlainSlerp
        self.smoothing_algo = algo if algo else PlainSlerp()

    def update_smoothing(self):
        """reference :256-267: integrate, smooth, resample at the frame rate -> self.times, self.stab_transform."""
        if getattr(self, "new_integrator", None) is not None:
            if getattr(self, "smoothing_algo", None) is None:
                self.set_smoothing_algo()
            self.new_integrator.integrate_all(use_acc=False)
            self.new_integrator.set_smoothing_algo(self.smoothing_algo)
            self.times, self.stab_transform = self.new_integrator.get_interpolated_stab_transform(start=0, interval=1 / self.fps)
            return True
        print("Orientations not calculated yet")
        return False

    def import_gyroflow_file(self, filename="file.gyroflow"):
        """reference :1676-1712: calibration + raw IMU from a Gyroflow data file; orientations are integrated from the
        gyro columns (the accelerometer columns are kept but unused: use_acc=False as in the reference's import)."""
        import json
        from scipy.spatial.transform import Rotation
        from .gyro_integrator import GyroIntegrator
        print("Loading Gyroflow data file")
        with open(filename, "r") as infile:
            try:
                gyroflow_data = json.load(infile)
                self.undistort.load_calibration_data(gyroflow_data["calibration_data"], True)
                self.video_rotate_code = gyroflow_data["video_rotate_code"]
                initial_orientation = Rotation.from_euler('zxy', [0, 0, np.pi / 2]).as_quat()
                initial_orientation[[0, 1, 2, 3]] = initial_orientation[[3, 0, 1, 2]]
                raw_imu = np.array(gyroflow_data["raw_imu"])
                self.gyro_data = raw_imu[:, [0, 1, 2, 3]]
                self.acc_data = raw_imu[:, [0, 4, 5, 6]] if gyroflow_data["sensor_DOF"] >= 6 else None
                self.integrator = GyroIntegrator(self.gyro_data, zero_out_time=False, initial_orientation=initial_orientation,
                                                 acc_data=self.acc_data)
                self.new_integrator = self.integrator
                self.new_integrator.integrate_all(use_acc=False)
                self.use_gyroflow_data_file = True
            except KeyError:
                print("Couldn't load gyroflow data file")
                return False
        return True

    def export_gyroflow_file(self, filename, out_size=(1920, 1080), smoothingFocus=2.0, videofile=""):
        """reference :1574-1672: the per-frame data plugins consume: [index, time, fcorr, w, x, y, z] rows rounded to 5
        decimals, min_fcorr, frame orientations, raw IMU and the calibration."""
        import json
        from datetime import date
        integ = getattr(self, "new_integrator", None)
        if integ is None or self.stab_transform is None:
            print("Unable to export a data file")
            return False
        data = {"title": "Gyroflow data file", "videofile": videofile,
                "calibration_data": self.undistort.get_calibration_data(),
                "video_rotate_code": getattr(self, "video_rotate_code", -1), "gyro_lpf_cutoff": getattr(self, "gyro_lpf_cutoff", -1),
                "date": str(date.today()),
                "video_info": {"orig_w": self.orig_dimension[0], "orig_h": self.orig_dimension[1], "process_w": self.width,
                               "process_h": self.width, "fps": self.fps, "num_frames": getattr(self.cap, "count", len(self.stab_transform))},
                "sensor_DOF": 6 if getattr(self, "acc_data", None) is not None else 3}
        raw = integ.gyro[:, :4] if getattr(self, "acc_data", None) is None else np.column_stack([integ.gyro[:, :4], self.acc_data[:, 1:4]])
        data["raw_imu"] = np.round(raw, 5).tolist()
        data["stab_summary"] = self.smoothing_algo.get_summary()
        t, o = integ.get_interpolated_orientations(start=0, interval=1 / self.fps)
        data["frame_orientation"] = np.hstack([np.arange(o.shape[0])[..., None], np.array(t)[..., None], o]).tolist()
        data["frame_orientation_filtered"] = []
        stab = np.array(self.stab_transform)
        fcorr, _ = AdaptiveZoom(fisheyeCalibrator=self.undistort, device=self.device).compute(
            quaternions=self.stab_transform, output_dim=out_size, fps=self.fps, smoothingFocus=smoothingFocus)
        data["min_fcorr"] = float(np.min(fcorr))
        rows = np.hstack([np.arange(stab.shape[0])[..., None], np.array(self.times)[..., None], np.asarray(fcorr, np.float64)[..., None], stab])
        data["stab_transform"] = np.round(rows, 5).tolist()
        with open(filename, 'w') as outfile:
            json.dump(data, outfile, indent=1, separators=(',', ': '))
        print("Finished exporting")
        return True

    def set_hyperlapse(self, hyperlapse_multiplier=1, hyperlapse_num_blended_frames=1):
        # frames |0|1|2|3|4|5|6|7|...  mult=4, num_blend=2  ->  |0+1|4+5|...
        self.hyperlapse_multiplier = hyperlapse_multiplier
        self.hyperlapse_num_blended_frames = min(hyperlapse_multiplier, hyperlapse_num_blended_frames)
        self.hyperlapse_skipped_frames = self.hyperlapse_multiplier - self.hyperlapse_num_blended_frames

    def set_map_func_scale(self, map_scale=0.9):
        self.map_func_scale = map_scale

    def map_function(self, frame_num, out_size=(1280, 720), regenerate=False):
        """Preview maps for one frame (reference :1205-1215): fixed-point (map1, map2) CUDA tensors."""
        if frame_num >= len(self.stab_transform):
            frame_num = len(self.stab_transform) - 1
            print("No more stabilization data. Using last frame")
        return self.undistort.get_maps(self.map_func_scale,
                                       new_img_dim=(self.orig_dimension[0], self.orig_dimension[1]),
                                       output_dim=out_size, update_new_K=False, quat=self.stab_transform[frame_num],
                                       focalCenter=None, original_stretched=True, device=self.device)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def border_from_bg_color(bg_color):
        """reference :1235-1244"""
        if bg_color == "REPLICATE":
            return cv.BORDER_REPLICATE, 0
        if bg_color == "HISTORY":
            return cv.BORDER_TRANSPARENT, 0
        return cv.BORDER_CONSTANT, [round(x * 255) for x in colors.to_rgb(bg_color)][::-1]

    def _fov_terms(self, out_size, smoothingFocus, tstart, tend):
        """fcorr[n], focalCenter[n] of the reference's AdaptiveZoom.compute call (stabilizer.py:1338-1342)."""
        n = len(self.stab_transform)
        if self.adaptive_zoom is not None:
            fcorr, focal = self.adaptive_zoom(self.stab_transform, out_size, self.fps, smoothingFocus, tstart, tend)
            return np.asarray(fcorr, np.float64), np.asarray(focal, np.float64)
        if smoothingFocus == -2:       # adaptive_zoom.py:152-153 discards the polygons: all ones, crop centre = calib centre
            cd = self.undistort.calib_dimension
            return np.ones(n), np.tile(np.array([cd[0] / 2, cd[1] / 2], np.float64), (n, 1))
        adaptZ = AdaptiveZoom(fisheyeCalibrator=self.undistort, device=self.device)
        fcorr, focal = adaptZ.compute(quaternions=self.stab_transform, output_dim=out_size, fps=self.fps,
                                      smoothingFocus=smoothingFocus, tstart=tstart, tend=tend)
        return np.asarray(fcorr, np.float64), np.asarray(focal, np.float64)

    def _xform(self, fov, frame_num, out_size, focal):
        return self.undistort.frame_xform(fov, output_dim=out_size, new_img_dim=self.orig_dimension,
                                          quat=self.stab_transform[frame_num], focalCenter=focal)

    # ------------------------------------------------------------------ render loop
    def schedule(self, first_pos, n_iter, first_iter=0, with_iter=False):
        """The reference loop's control flow (stabilizer.py:1378-1411, 1431-1438, 1464) without any pixel work.

        Reads n_iter frames starting at source position first_pos and yields, for every successfully read frame,
        (frame_num, frame, phase, remap, emit): phase = (i-1) % hyperlapse_multiplier with i the reference's loop
        counter, remap = the frame is inside the blended window, emit = an output frame is written after it
        (with_iter=True appends i-1 itself).  Stops on exhausted stab_transform or after more than five consecutive
        failed reads.  Pure host logic."""
        mult, nblend = self.hyperlapse_multiplier, self.hyperlapse_num_blended_frames
        self.cap.set_pos(first_pos)
        num_not_success, num_not_success_lim = 0, 5
        for k in range(first_iter, first_iter + n_iter):          # k = i - 1
            frame_num = int(self.cap.get_pos())
            success, frame = self.cap.read()
            if success:
                num_not_success = 0
            elif num_not_success >= num_not_success_lim:
                print("Unable to read multiple frames")
                return
            else:
                num_not_success += 1
            if frame_num >= len(self.stab_transform):
                print("No more stabilization data. Stopping render here.")
                return
            if not success:
                continue
            phase = k % mult
            ev = (frame_num, frame, phase, phase < nblend, (phase - nblend + 1) % mult == 0)
            yield ev + (k,) if with_iter else ev

    VALID_EXTENSIONS = ["webm", "flv", "vob", "ogv", "ogg", "drc", "gif", "mng", "avi", "mts", "m2ts", "ts", "mov", "qt", "yuv",
                        "rmvb", "viv", "asf", "amv", "mp4", "m4p", "m4v", "mpg", "mpeg", "m2v", "svi", "3gp", "3g2", "mxf", "roq",
                        "nsv", "flv"]

    @staticmethod
    def default_batch(frame_bytes):
        """Frames per launch of the batched path: the fused kernel's launch cost is about 68 us + 19 us per 4K frame
        (DESIGN.md K1), so launches of 64 4K frames (1.6 GB) run within 6 % of the steady rate; smaller frames get
        proportionally more per launch, capped at 256."""
        return int(min(256, max(8, (64 * 3840 * 2160 * 3) // max(int(frame_bytes), 1))))

    def renderfile(self, starttime, stoptime, outpath="Stabilized.mp4", out_size=(1920, 1080), split_screen=False,
                   bitrate_mbits=20, display_preview=False, scale=1, vcodec="libx264", vprofile="main", pix_fmt="",
                   debug_text=False, custom_ffmpeg="", smoothingFocus=4.0, fov_scale=1.0, bg_color="#000000", audio=True,
                   viewer_thread=None, *, batch=None, pipeline_depth=3, sequential=False, rank=0, world=1):
        """Render [starttime, stoptime) through the fused undistort+rotate kernel (reference: stabilizer.py:1220-1531, same
        positional and keyword arguments).  Returns the number of frames written (the reference returns None).

        `outpath` is a frame sink (anything with write(frame) / close(): ListSink, NullSink, RawVideoSink, an encoder
        pipe) or a path: encoding is out of scope here, so a path receives the frames as headerless raw video (what the
        reference pipes into ffmpeg through WriteGear, :1326) after the reference's file-extension check (:1230-1233).  The
        encoder options (bitrate_mbits, vcodec, vprofile, pix_fmt, custom_ffmpeg -- never eval'd --, audio) are accepted and
        ignored; split_screen and scale are dead in the reference too (:1470, :1415) and ignored alike; debug_text draws
        with cv2 and is not available: the call prints that and returns, like the reference's own early exits.  Nothing
        raises out of the loop: a failed write prints and stops the render (:1486-1491).

        Keyword-only extensions: `sequential=True` issues the reference's per-frame sequence get_maps -> remap -> blend with a
        blocking download per frame; otherwise frames are batched (`batch` per launch, default by frame size and clip length)
        through the pinned multi-stream pipeline and the fused kernels.  bg_color="HISTORY" (BORDER_TRANSPARENT: every frame
        is drawn over the previous output, :1240-1241,1442) is batched too: the frames of a batch are chained on the GPU --
        one fused launch per frame into the running output frame, on one stream -- so only the transfers are pipelined.
        With world > 1 this rank renders only its contiguous shard of the loop iterations (frame_shard); concatenating the
        ranks' outputs in rank order gives the single-GPU result (render_sharded() does that for a whole node)."""
        if isinstance(outpath, str) and outpath.split(".")[-1].lower() not in self.VALID_EXTENSIONS:
            print(f"{outpath} does not have a valid file extension")
            return
        if debug_text:
            print("debug_text overlays are drawn with cv2.rectangle/cv2.putText in the reference: not available here")
            return
        borderMode, borderValue = self.border_from_bg_color(bg_color)
        if world > 1 and borderMode == cv.BORDER_TRANSPARENT:
            print('bg_color="HISTORY" makes every frame depend on the previous output: it does not shard')
            return
        if (vcodec, vprofile, pix_fmt, custom_ffmpeg, bitrate_mbits, audio) != ("libx264", "main", "", "", 20, True):
            print("Note: encoder options are ignored (frames go to the sink unencoded)")
        if isinstance(outpath, str):
            from .frameio import RawVideoSink
            sink = RawVideoSink(outpath)
        else:
            sink = outpath
        num_frames = int((stoptime - starttime) * self.fps)
        tstart = int(starttime * self.fps)
        tend = tstart + num_frames
        print("Starting to compute optimal Fov")
        fcorr, focalCenter = self._fov_terms(out_size, smoothingFocus, tstart, tend)
        print("Done computing optimal Fov")

        mult, nblend = self.hyperlapse_multiplier, self.hyperlapse_num_blended_frames
        rot = getattr(self, "video_rotate_code", -1)           # import_gyroflow_file may have set it (:1688)
        rot = -1 if rot is None else int(rot)
        if rot not in (-1, cv.ROTATE_90_CLOCKWISE, cv.ROTATE_180, cv.ROTATE_90_COUNTERCLOCKWISE):
            print("video_rotate_code %r is not a cv2.rotate code" % (rot,))
            return
        hist = borderMode == cv.BORDER_TRANSPARENT and not sequential       # HISTORY: frames chained on the GPU inside a batch
        lo, hi = frame_shard(max(num_frames - 1, 0), rank, world, mult)     # the loop runs i = 1 .. num_frames-1
        stretched = self.undistort.image_is_stretched() and nblend > 1 and not hist     # the fused blend kernel has no stretched-preset form

        old_map_enable_setting = False
        if viewer_thread is not None and display_preview:
            old_map_enable_setting = getattr(viewer_thread, "map_function_enable", False)
            viewer_thread.map_function_enable = False
        if display_preview and viewer_thread is None:
            print("Note: no preview window here (cv2.imshow); pass viewer_thread to receive preview frames")

        state = dict(written=0, abort=False)

        def emit(frames):
            for f in frames:
                if state["abort"]:
                    return
                try:
                    sink.write(f)
                except Exception as e:
                    print("Failed to write frame. Aborting render")
                    print(e)
                    state["abort"] = True
                    return
                state["written"] += 1
                if display_preview and viewer_thread is not None:
                    try:
                        fp = f
                        if f.shape[1] > 1280:        # cv2.resize(..., INTER_NEAREST): source index = floor(dst * scale)
                            ph = int(f.shape[0] * 1280 / f.shape[1])
                            ys = np.minimum((np.arange(ph) * (f.shape[0] / ph)).astype(np.int64), f.shape[0] - 1)
                            xs = np.minimum((np.arange(1280) * (f.shape[1] / 1280)).astype(np.int64), f.shape[1] - 1)
                            fp = f[ys][:, xs]
                        viewer_thread.frame = np.array(fp, copy=True)
                        viewer_thread.update_once = True
                    except Exception as e:
                        print("Failed to display preview")
                        print(e)

        dtype = None
        pipe = None
        seq = None                              # state of the sequential path: frame_out, frame_temp (float64)

        def seq_state():
            nonlocal seq
            if seq is None:
                seq = dict(frame_out=None, frame_temp=torch.zeros((out_size[1], out_size[0], 3), dtype=torch.float64, device=self.device)
                           if nblend > 1 else None)
            return seq

        def seq_frame(frame, frame_num, phase, do_remap, do_emit):
            """One iteration of the reference loop body (:1423-1468 and :1487) with its own kernels."""
            st = seq_state()
            reset = phase == 0 and nblend > 1          # frame_temp * 0.0 at the start of a hyperlapse group
            if do_remap:
                m1, m2 = self.undistort.get_maps(fov_scale * fcorr[frame_num], new_img_dim=self.orig_dimension, output_dim=out_size,
                                                 update_new_K=False, quat=self.stab_transform[frame_num],
                                                 focalCenter=focalCenter[frame_num], device=self.device)
                src = frame.to(self.device) if isinstance(frame, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame)).to(self.device)
                if rot != -1:
                    src = cv.rotate(src, rot)                                  # :1385-1386
                st["frame_out"] = cv.remap(src, m1, m2, cv.INTER_LINEAR, dst=st["frame_out"], borderMode=borderMode,
                                           borderValue=borderValue)
                if nblend > 1:
                    cv.blend_accumulate(st["frame_out"], st["frame_temp"], 1 / nblend, reset)
            elif reset:
                st["frame_temp"].mul_(0.0)
            if do_emit and st["frame_out"] is not None:
                if nblend > 1:
                    st["frame_out"] = cv.blend_finalize(st["frame_temp"], dtype)
                emit([st["frame_out"].cpu().numpy()])

        # ---- batched path state
        cur = dict(slot=None, n=0, fnums=[], meta=[])    # the batch being staged (meta: (phase, emit) per frame, HISTORY only)
        group = []                              # (frame, frame_num, phase, do_emit) of the hyperlapse group being collected
        group_id = None
        prev_clean = None                       # frames of the last group that went through the fused blend kernel
        seq_valid = True                        # the sequential state reflects everything rendered so far

        def xforms(fnums):
            fn = np.asarray(fnums, np.int64)
            return self.undistort.frame_xforms(fov_scale * fcorr[fn], self.stab_transform[fn], output_dim=out_size,
                                               new_img_dim=self.orig_dimension, focalCenters=focalCenter[fn])

        def flush_batch():
            n = cur["n"]
            if n == 0:
                return
            slot, fnums, meta = cur["slot"], cur["fnums"], cur["meta"]
            cur.update(slot=None, n=0, fnums=[], meta=[])
            n_groups = n // nblend

            def launch_history(din, dout, dxf):
                """The loop body of :1431-1468 for every frame of the batch, in order, on the compute stream: frame n is remapped
                OVER the running output frame (the pixels its source does not cover keep what frame n - 1 left there), blended
                if a hyperlapse is on, and copied to the batch's output when the reference would write it."""
                st = seq_state()
                if st["frame_out"] is None:         # cv2.remap creates dst on the first call; zero-filled here and in cv.remap
                    st["frame_out"] = torch.zeros((out_size[1], out_size[0], 3), dtype=din.dtype, device=self.device)
                k = 0
                for i, (phase, do_emit) in enumerate(meta):
                    cv.undistort_rotate(din[i:i + 1], self.undistort.D, dxf[i:i + 1], out_size, dst=st["frame_out"][None],
                                        borderMode=borderMode, borderValue=borderValue, map_scale=self.undistort.map_scale(), rotate=rot)
                    if nblend > 1:
                        cv.blend_accumulate(st["frame_out"], st["frame_temp"], 1 / nblend, phase == 0)
                    if do_emit:
                        if nblend > 1:
                            cv.blend_finalize(st["frame_temp"], dtype, out=st["frame_out"])
                        dout[k].copy_(st["frame_out"])
                        k += 1
                return k

            def launch(din, dout, dxf):
                if hist:
                    return launch_history(din, dout, dxf)
                if nblend == 1:
                    cv.undistort_rotate(din, self.undistort.D, dxf, out_size, dst=dout[:n], borderMode=borderMode, borderValue=borderValue,
                                        map_scale=self.undistort.map_scale(), rotate=rot)      # the rotation rides in the tap addresses
                else:
                    if rot != -1:
                        din = cv.rotate(din, rot)
                    cv.undistort_rotate_blend(din, self.undistort.D, dxf, out_size, nblend, nblend, dst=dout[:n_groups],
                                              borderMode=borderMode, borderValue=borderValue)
                return n_groups
            pipe.commit(slot, n, launch, xforms(fnums))

        def stage(frame, frame_num, meta=None):
            nonlocal pipe
            if pipe is None:
                fbytes = frame.nbytes if hasattr(frame, "nbytes") else frame.numel() * frame.element_size()
                if batch:
                    b = int(batch)
                else:       # no more ring than the clip can fill: a 30-frame render must not page-lock a 64-frame ring three times
                    remapped = -(-(hi - lo) // mult) * nblend
                    b = max(nblend, min(self.default_batch(fbytes), -(-remapped // max(int(pipeline_depth), 1))))
                out_batch = b if (nblend == 1 or hist) else max(1, b // nblend)
                in_batch = b if hist else out_batch * nblend
                # the pinned ring is expensive to allocate (page-locking gigabytes): keep it between renders of the same shape.  A
                # ring sized by default may be larger than this render needs; an explicit `batch` is taken literally.
                shape_key = (frame.shape[0], frame.shape[1], out_size[1], out_size[0], dtype, pipeline_depth, nblend, hist)
                cached, ckey = getattr(self, "_pipe", None), getattr(self, "_pipe_key", None)
                if cached is not None and ckey is not None and ckey[0] == shape_key and \
                        ((batch and (cached.batch, cached.out_batch) == (in_batch, out_batch)) or
                         (not batch and ckey[1] and cached.batch >= in_batch)):
                    pipe = cached
                    pipe.deliver = emit
                else:
                    self._pipe = self._pipe_key = None
                    pipe = BatchPipeline(self.device, (frame.shape[0], frame.shape[1]), (out_size[1], out_size[0]), dtype,
                                         in_batch, emit, depth=pipeline_depth, out_batch=out_batch)
                    self._pipe, self._pipe_key = pipe, (shape_key, not batch)
            if cur["slot"] is None:
                cur["slot"] = pipe.acquire()
            pipe.stage(cur["slot"], cur["n"], frame)
            cur["n"] += 1
            cur["fnums"].append(frame_num)
            if meta is not None:
                cur["meta"].append(meta)
            if cur["n"] >= pipe.batch:
                flush_batch()

        def close_group():
            """A finished hyperlapse group (nblend > 1): whole groups -- phases 0 .. nblend-1 all read -- go to the fused blend
            kernel; a group with a failed read inside takes the reference's frame-by-frame sequence instead (there a
            missing first frame also means a missing reset, so the accumulator of the previous group carries over)."""
            nonlocal group, prev_clean, seq_valid
            g, group = group, []
            if not g:
                return
            if [e[2] for e in g] == list(range(nblend)):
                for frame, frame_num, _, _ in g:
                    stage(frame, frame_num)
                prev_clean, seq_valid = [(e[0], e[1]) for e in g], False
                return
            flush_batch()
            if pipe is not None:
                pipe.flush()                    # keep the output order
            if not seq_valid and prev_clean is not None and g[0][2] != 0:
                for b, (frame, frame_num) in enumerate(prev_clean):      # rebuild frame_temp as the previous group left it
                    seq_frame(frame, frame_num, b, True, False)
            for frame, frame_num, phase, do_emit in g:
                seq_frame(frame, frame_num, phase, True, do_emit)
            prev_clean, seq_valid = None, True

        finished = False
        try:
            for frame_num, frame, phase, do_remap, do_emit, k in self.schedule(tstart + lo, hi - lo, first_iter=lo, with_iter=True):
                if state["abort"]:
                    break
                if dtype is None:
                    dtype = torch.uint8 if str(frame.dtype).endswith("uint8") else torch.uint16
                if sequential or stretched:
                    seq_frame(frame, frame_num, phase, do_remap, do_emit)
                    continue
                if hist:
                    if do_remap:
                        stage(frame, frame_num, (phase, do_emit))
                    continue
                if nblend == 1:
                    if do_remap:
                        stage(frame, frame_num)
                    continue
                if group_id is not None and k // mult != group_id:
                    close_group()
                group_id = k // mult
                if do_remap:
                    group.append((frame, frame_num, phase, do_emit))
            if not state["abort"]:
                if nblend > 1 and not (sequential or stretched or hist):
                    # an unfinished trailing group emits nothing (as in the reference): only a group whose last blended frame
                    # was read is closed
                    if group and group[-1][3]:
                        close_group()
                flush_batch()
                if pipe is not None:
                    pipe.flush()
                finished = True
        except KeyboardInterrupt:
            print("terminating render")
        finally:
            # A render that stops early (failed write, Ctrl-C, an exception from the source) leaves batches in flight in the
            # cached ring: retire them without delivering, or the next render on this Stabilizer would write them into its sink.
            if pipe is not None and not finished:
                pipe.abort()
        print("Render finished")
        sink.close()
        if viewer_thread is not None:
            viewer_thread.map_function_enable = old_map_enable_setting
        return state["written"]


    def release_pipeline(self):
        """Drop the pinned ring renderfile keeps between renders of the same shape (several GB of page-locked host memory and as
        much HBM for 4K batches); the next render allocates a new one."""
        self._pipe = self._pipe_key = None

    close = release_pipeline


class OnlyUndistort:
    """Lens undistortion without stabilisation: mirror of the reference's OnlyUndistort (stabilizer.py:1718-1844).

    The reference builds static maps once (get_maps(fov_scale, new_img_dim=(w, h)), :1729), remaps every frame with
    BORDER_CONSTANT (:1825) and crops the centre out_size window (:1792,1828).  Here the static transform is handed to the
    fused kernel for every frame of a batch.  scale != 1 needs cv2.resize of the maps and the frame (:1800-1801,1823):
    out of scope, rejected.  Deviation: the reference loops forever on a source that keeps failing; this stops when the
    source is exhausted."""

    def __init__(self, source, calibrationfile, width, height, fps, fov_scale=1.5, device=None):
        self.undistort_fov_scale = fov_scale
        self.cap = source
        self.width, self.height, self.fps = int(width), int(height), fps
        if isinstance(calibrationfile, FisheyeCalibrator):
            self.undistort = calibrationfile
        else:
            self.undistort = FisheyeCalibrator()
            self.undistort.load_calibration_json(calibrationfile, True)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.map1, self.map2 = self.undistort.get_maps(self.undistort_fov_scale, new_img_dim=(self.width, self.height),
                                                       device=self.device)

    def renderfile(self, starttime, stoptime, sink, out_size=(1920, 1080), scale=1, batch=8, pipeline_depth=2):
        if scale != 1:
            raise NotImplementedError("scale != 1 resizes the maps with cv2.resize(INTER_CUBIC): out of scope")
        crop = (int(scale * (self.width - out_size[0]) / 2), int(scale * (self.height - out_size[1]) / 2))
        if crop[0] < 0 or crop[1] < 0:
            raise ValueError("out_size larger than the frame: numpy's negative slice start would wrap, as in the reference")
        self.cap.set_pos(int(starttime * self.fps))
        num_frames = int((stoptime - starttime) * self.fps)
        map_h, map_w = int(self.map2.shape[0]), int(self.map2.shape[1])       # = calib_dimension (get_maps default output_dim)
        xf1 = self.undistort.frame_xform(self.undistort_fov_scale, output_dim=None, new_img_dim=(self.width, self.height))
        written = [0]

        def emit(frames):
            for f in frames:
                sink.write(f[crop[1]:crop[1] + out_size[1] * scale, crop[0]:crop[0] + out_size[0] * scale])
                written[0] += 1

        pipe, pend, i = None, [], 0

        def flush():
            nonlocal pend
            if not pend:
                return
            n = len(pend)
            pipe.submit(pend, lambda din, dout, dxf: (cv.undistort_rotate(din, self.undistort.D, dxf, (map_w, map_h), dst=dout[:n],
                                                                          borderMode=cv.BORDER_CONSTANT, map_scale=self.undistort.map_scale()), n)[1],
                        xf=np.tile(xf1, (n, 1)))
            pend = []

        while True:
            success, frame = self.cap.read()
            if success:
                i += 1
            else:
                break
            if i > num_frames:
                break
            if pipe is None:
                dtype = torch.uint8 if frame.dtype == np.uint8 else torch.uint16
                pipe = BatchPipeline(self.device, (frame.shape[0], frame.shape[1]), (map_h, map_w), dtype, batch, emit,
                                     depth=pipeline_depth)
            pend.append(frame)
            if len(pend) >= batch:
                flush()
        if pipe is not None:
            flush()
            pipe.flush()
        sink.close()
        return written[0]
