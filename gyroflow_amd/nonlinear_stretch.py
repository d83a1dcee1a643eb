"""NonlinearStretch: 4:3 -> 16:9 "superview" stretch, bicubic remap on MI355X.

Host-side mirror of /root/reference/nonlinear_stretch.py:6-139.  recompute_maps() evaluates the reference's
closed form in float64 numpy and casts to float32 (one-shot per configuration, :58-100); apply_stretch() is
cv2.remap(INTER_CUBIC) (:130) executed by libgfremap's bicubic kernel.
"""
import numpy as np
import torch

from . import cv


class NonlinearStretch:
    def __init__(self, in_size=(12, 9), out_size=(16, 9), safe_area=0, expo=2):
        self.in_size = in_size
        self.out_size = out_size
        self.safe_area = safe_area
        self.expo = expo
        self.map1 = np.zeros((out_size[1], out_size[0]))   # x coordinates
        self.map2 = np.zeros((out_size[1], out_size[0]))   # y coordinates
        self._dev_maps = None

    def set_safe_area(self, safe_area):
        """Untouched centre fraction (0..1), clamped below 1."""
        self.safe_area = min(safe_area, 0.999)

    def set_in_size(self, in_size):
        self.in_size = in_size

    def set_expo(self, expo=2):
        """1 = linear stretch, 2 = superview-like, larger = more non-linear."""
        self.expo = expo

    def set_out_size(self, out_size):
        self.out_size = out_size
        self.map1 = np.zeros((out_size[1], out_size[0]))
        self.map2 = np.zeros((out_size[1], out_size[0]))
        self._dev_maps = None

    def recompute_maps(self):
        out_w, out_h = self.out_size
        vertical_scale = out_h / self.in_size[1]                       # match heights
        pillar_width = (out_w / vertical_scale - self.in_size[0]) / 2  # unstretched side pillar, source px

        cols = np.tile(np.arange(out_w), (out_h, 1))
        map_x = cols / vertical_scale - pillar_width - 0.5             # centred source x, half-pixel fix

        nx = (cols / out_w - 0.5) * 2                                  # -1 <= nx < 1
        sign = np.zeros((out_h, out_w))
        sign[nx < 0] = -1
        sign[nx >= 0] = 1
        with np.errstate(invalid="ignore"):
            offset = ((abs(nx) - self.safe_area) / (1 - self.safe_area)) ** self.expo
        offset[abs(nx) < self.safe_area] = 0                           # no offset inside the safe area
        offset = np.multiply(offset, sign) * pillar_width
        map_x = map_x - offset

        rows = np.tile(np.vstack(np.arange(out_h)), (1, out_w))
        map_y = rows / vertical_scale - 0.5

        self.map1 = map_x.astype('float32')
        self.map2 = map_y.astype('float32')
        self._dev_maps = None

    def _maps_on(self, device):
        if self._dev_maps is None or self._dev_maps[0].device != device:
            self._dev_maps = (torch.from_numpy(np.ascontiguousarray(self.map1.astype('float32'))).to(device),
                              torch.from_numpy(np.ascontiguousarray(self.map2.astype('float32'))).to(device))
        return self._dev_maps

    def apply_stretch_batch(self, frames):
        """Stretch CUDA uint8 frames [N,H,W,3] (or one [H,W,3]) in one launch.  recompute_maps() makes map1 a function of
        the column and map2 a function of the row, so the kernel reads one row / one column of them instead of the
        two full-size float32 maps; the result equals cv2.remap(img, map1, map2, cv2.INTER_CUBIC) per frame."""
        m1, m2 = self._maps_on(frames.device)
        return cv.superview_cubic(frames, m1[0, :], m2[:, 0])

    def apply_stretch(self, img, show_protected=False):
        """Stretch one uint8 HxWx3 image (numpy -> numpy, CUDA tensor -> CUDA tensor)."""
        if isinstance(img, torch.Tensor) and img.is_cuda:
            m1, m2 = self._maps_on(img.device)
        else:
            m1, m2 = self._maps_on(torch.device("cuda", torch.cuda.current_device()))
        out_img = cv.remap(img, m1, m2, cv.INTER_CUBIC)
        if show_protected:
            # two vertical marker lines at the edges of the safe area (cv2.line, thickness 2, colour (255,255,0))
            midpoint = out_img.shape[1] / 2
            safe_dist = self.safe_area * out_img.shape[1] / 2
            colour = (255, 255, 0)
            for line in (int(midpoint + safe_dist), int(midpoint - safe_dist)):
                lo, hi = max(line - 1, 0), min(line + 1, out_img.shape[1])   # approx. 2 px wide
                for c in range(3):
                    out_img[:, lo:hi, c] = colour[c]
        return out_img
