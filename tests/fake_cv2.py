"""A stand-in `cv2` for tests ONLY: the handful of OpenCV calls WebvidDatasetV2's decode branch makes (data/v2v_datasets.py:188-213, 252-256),
backed by oracle/frontend_oracle.py's restatement of OpenCV 4.x's 8-bit algorithms and by "videos" that are .npy arrays [T,H,W,3] on disk.
There is no OpenCV in this image, so the `video_reader: opencv` branches of v2v_amd/datasets.py (read_video, _probe_size, read_video_gpu)
never ran before round 5; with this module injected as sys.modules['cv2'] they do, and their output is compared with the frame_source path
(which the other tests pin) -- it tests the CONTROL FLOW of the branch (seek, read loop, cvtColor before the crop, resize to need_w x need_h,
flip after the resize, expand_dims, release), not OpenCV itself."""
import numpy as np

from oracle import frontend_oracle as FO

CAP_PROP_POS_FRAMES, CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT = 1, 3, 4
COLOR_BGR2GRAY, COLOR_GRAY2BGR = 6, 8
INTER_LINEAR = 1
calls = []                                                  # (name, args) log the tests inspect


class VideoCapture:
    def __init__(self, path):
        self.frames = np.load(path, mmap_mode="r")         # [T,H,W,3] uint8
        self.pos = 0
        self.open = True
        calls.append(("VideoCapture", path))

    def get(self, prop):
        return {CAP_PROP_FRAME_WIDTH: self.frames.shape[2], CAP_PROP_FRAME_HEIGHT: self.frames.shape[1], CAP_PROP_POS_FRAMES: self.pos}[prop]

    def set(self, prop, value):
        assert prop == CAP_PROP_POS_FRAMES
        self.pos = int(value)
        calls.append(("set_pos", int(value)))

    def read(self):
        assert self.open
        if self.pos >= self.frames.shape[0]:
            return False, None
        f = np.array(self.frames[self.pos])
        self.pos += 1
        return True, f

    def release(self):
        self.open = False
        calls.append(("release", None))


def cvtColor(img, code):
    assert code == COLOR_BGR2GRAY and img.ndim == 3 and img.shape[2] == 3
    return FO.cv_bgr2gray_u8(img, "cv4")


def resize(img, dsize, interpolation=INTER_LINEAR):
    assert interpolation == INTER_LINEAR
    dw, dh = dsize
    return FO.cv_resize_linear_u8(img, dw, dh)              # a [H,W] image stays 2-D: cv2.resize eats a trailing 1-channel axis (the reference relies on it)


def flip(img, code):
    assert code == 1
    return np.ascontiguousarray(img[:, ::-1])
