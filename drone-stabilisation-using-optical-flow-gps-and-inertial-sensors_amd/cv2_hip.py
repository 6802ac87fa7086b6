"""cv2_hip.py — the cv2 calls on the reference's hot path, with OpenCV's names, signatures and array
conventions, executed by the HIP kernels of libofk.so.  `import cv2_hip as cv2` in the reference's scripts
covers: cvtColor(COLOR_BGR2GRAY), goodFeaturesToTrack, calcOpticalFlowPyrLK, KalmanFilter, TERM_CRITERIA_*.

Reference call sites: of_module.py:40,44,63-76,80,86,88,122,152; velocity_measurment_node:113,120,133,163;
evaluate_exp.py:65,66,85,98,106; of_library.py:236,238,248,249.
Semantics are those of oracle/image_oracle.c (OpenCV's published algorithms with exact integer window sums).
"""
import numpy as np

try:
    from . import ofk
except ImportError:
    import ofk

COLOR_BGR2GRAY = 6
TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_MAX_ITER = 1
TERM_CRITERIA_EPS = 2


def cvtColor(src, code):
    if code != COLOR_BGR2GRAY:
        raise NotImplementedError("only COLOR_BGR2GRAY is on the hot path")
    src = np.asarray(src)
    if src.ndim != 3 or src.shape[2] != 3 or src.dtype != np.uint8:
        raise ValueError("cvtColor(BGR2GRAY) expects an HxWx3 uint8 image")
    h, w = src.shape[:2]
    return ofk.default_context(w, h).gray_bgr8(src)


IMREAD_UNCHANGED, IMREAD_GRAYSCALE, IMREAD_COLOR = -1, 0, 1


def imdecode(buf, flags=IMREAD_COLOR):
    """cv2.imdecode for baseline JPEG streams (what cv_bridge.compressed_imgmsg_to_cv2 calls for the reference's CompressedImage
    callback, velocity_measurment_node.py:112), decoded on the GPU.  IMREAD_COLOR: HxWx3 BGR; IMREAD_UNCHANGED: HxW for gray
    streams, HxWx3 otherwise.  Returns None - like OpenCV - if the buffer is not a stream the decoder supports."""
    if flags not in (IMREAD_COLOR, IMREAD_UNCHANGED):
        raise NotImplementedError("imdecode: IMREAD_COLOR and IMREAD_UNCHANGED only (gray conversion is cvtColor's job on this path)")
    data = np.asarray(buf, dtype=np.uint8).tobytes() if not isinstance(buf, (bytes, bytearray, memoryview)) else bytes(buf)
    try:
        h, w, ncomp = ofk.jpeg_info(data)
        if h > ofk.MAX_DIM or w > ofk.MAX_DIM:      # a corrupt SOF header can claim 65535 x 65535: not a frame, no context for it
            return None
        img = ofk.default_context(w, h).jpeg_decode([data])[0]
    except ofk.OfkError:                            # unsupported / truncated stream, or no memory for a frame this size
        return None
    return np.ascontiguousarray(img[:, :, 0]) if (ncomp == 1 and flags == IMREAD_UNCHANGED) else img


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, corners=None, mask=None, blockSize=3,
                        useHarrisDetector=False, k=0.04):
    if useHarrisDetector:
        raise NotImplementedError("the reference only uses the min-eigenvalue (Shi-Tomasi) detector")
    image = np.asarray(image)
    if image.ndim != 2 or image.dtype != np.uint8:
        raise ValueError("goodFeaturesToTrack expects an HxW uint8 image")
    h, w = image.shape
    if maxCorners <= 0:
        maxCorners = 4096
    ctx = ofk.default_context(w, h, min_pts=int(maxCorners))
    pts = ctx.good_features(image, int(maxCorners), float(qualityLevel), float(minDistance), int(blockSize), mask=mask)
    return pts if len(pts) else None           # OpenCV's Python binding returns None when nothing is found


def _criteria(criteria):
    typ, cnt, eps = criteria
    if not typ & TERM_CRITERIA_COUNT:
        cnt = 30
    if not typ & TERM_CRITERIA_EPS:
        eps = 0.01
    return int(cnt), float(eps)


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, status=None, err=None, winSize=(21, 21), maxLevel=3,
                         criteria=(TERM_CRITERIA_COUNT | TERM_CRITERIA_EPS, 30, 0.01), flags=0, minEigThreshold=1e-4):
    if flags:
        raise NotImplementedError("OPTFLOW_USE_INITIAL_FLOW / LK_GET_MIN_EIGENVALS are not used by the reference")
    if winSize[0] != winSize[1]:
        raise NotImplementedError("square windows only (the reference uses (15,15))")
    prevImg = np.asarray(prevImg); nextImg = np.asarray(nextImg)
    if prevImg.ndim != 2 or prevImg.dtype != np.uint8 or nextImg.shape != prevImg.shape or nextImg.dtype != np.uint8:
        raise ValueError("calcOpticalFlowPyrLK expects two HxW uint8 images of equal size")
    h, w = prevImg.shape
    pts = np.asarray(prevPts, np.float32).reshape(-1, 2)
    cnt, eps = _criteria(criteria)
    ctx = ofk.default_context(w, h, min_pts=max(1, len(pts)), min_level=int(maxLevel))
    return ctx.lk_pyr(prevImg, nextImg, pts, win=int(winSize[0]), max_level=int(maxLevel), max_count=cnt, eps=eps,
                      min_eig_thr=float(minEigThreshold))


class KalmanFilter:
    """cv2.KalmanFilter(dynamParams, measureParams, controlParams) — predict()/correct() run ofk_kf_predict_update.
    Matrices are float64 (the reference assigns float64 numpy arrays, of_module.py:65-76)."""

    def __init__(self, dynamParams, measureParams, controlParams=0, type=None):
        ns, nm, nc = int(dynamParams), int(measureParams), int(controlParams)
        if not (1 <= ns <= 6 and 1 <= nm <= 6 and 0 <= nc <= 6):
            raise ValueError("state/measurement/control sizes up to 6 are supported")
        self._ns, self._nm, self._nc = ns, nm, nc
        self.transitionMatrix = np.eye(ns)
        self.controlMatrix = np.zeros((ns, nc)) if nc else None
        self.measurementMatrix = np.zeros((nm, ns))
        self.processNoiseCov = np.eye(ns)
        self.measurementNoiseCov = np.eye(nm)
        self.statePre = np.zeros((ns, 1)); self.statePost = np.zeros((ns, 1))
        self.errorCovPre = np.zeros((ns, ns)); self.errorCovPost = np.zeros((ns, ns))
        self.gain = np.zeros((ns, nm))

    def _mats(self):
        ns, nm = self._ns, self._nm
        return (np.asarray(self.transitionMatrix, np.float64).reshape(ns, ns),
                np.asarray(self.measurementMatrix, np.float64).reshape(nm, ns),
                np.asarray(self.processNoiseCov, np.float64).reshape(ns, ns),
                np.asarray(self.measurementNoiseCov, np.float64).reshape(nm, nm))

    def predict(self, control=None):
        F, H, Q, R = self._mats()
        ctx = ofk.default_context()
        Bm = u = None
        if control is not None and self.controlMatrix is not None and np.size(self.controlMatrix):
            nc = np.size(control)
            Bm = np.asarray(self.controlMatrix, np.float64).reshape(self._ns, nc); u = np.asarray(control, np.float64).reshape(nc)
        x, P = ctx.kf_predict_update(F, H, Q, R, np.asarray(self.statePost, np.float64).reshape(self._ns),
                                     np.asarray(self.errorCovPost, np.float64).reshape(self._ns, self._ns), B=Bm, u=u, z=None,
                                     do_predict=True)
        self.statePre = x.reshape(-1, 1).copy(); self.errorCovPre = P.copy()
        self.statePost = self.statePre.copy(); self.errorCovPost = self.errorCovPre.copy()
        return self.statePre

    def correct(self, measurement):
        F, H, Q, R = self._mats()
        ctx = ofk.default_context()
        x, P = ctx.kf_predict_update(F, H, Q, R, np.asarray(self.statePre, np.float64).reshape(self._ns),
                                     np.asarray(self.errorCovPre, np.float64).reshape(self._ns, self._ns),
                                     z=np.asarray(measurement, np.float64).reshape(self._nm), do_predict=False)
        self.statePost = x.reshape(-1, 1).copy(); self.errorCovPost = P.copy()
        return self.statePost
