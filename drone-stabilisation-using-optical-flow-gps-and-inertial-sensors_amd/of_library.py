"""of_library.py — drop-in for the reference's of_library.py (imported as `of`: velocity_measurment_node:7,
of_module.py:3).  Same function names and positional signatures; array math runs on the HIP kernels where the
reference runs a per-point Python loop or an OpenCV call.

Half of the reference module cannot execute as shipped (NameErrors / py2-only calls); those
functions keep their signature and implement the evidently intended behaviour — each docstring says so.
"""
import numpy as np

try:
    from . import ofk, cv2_hip as cv2
except ImportError:
    import ofk
    import cv2_hip as cv2


def visualize(image, mask, newpos, oldpos, frame_name="visualization", marker=[0, 0, 255], ):
    """of_library.py:19-27 draws trails with cv2.line/circle and cv2.imshow.  Display code is outside the hot
    path; this draws the same markers into `image`/`mask` with numpy and returns the blended frame instead of
    opening a window."""
    image = np.asarray(image); mask = np.asarray(mask)
    col = np.asarray(marker, image.dtype)
    for new, old in zip(newpos, oldpos):
        a, b = np.ravel(new)[:2]; c, d = np.ravel(old)[:2]
        n = int(max(abs(a - c), abs(b - d), 1))
        for s in np.linspace(0.0, 1.0, n + 1):
            x = int(round(a + (c - a) * s)); y = int(round(b + (d - b) * s))
            if 0 <= y < mask.shape[0] and 0 <= x < mask.shape[1]:
                mask[max(0, y - 1):y + 1, max(0, x - 1):x + 1] = col
        _disc(image, int(round(a)), int(round(b)), 5, col)
    return np.clip(image.astype(np.int32) + mask.astype(np.int32), 0, 255).astype(image.dtype)


def pix_trans(img_dim):
    """of_library.py:31-43."""
    if img_dim[0] % 2 == 0:
        trans_x = img_dim[0] / 2
    else:
        trans_x = (img_dim[0] + 1) / 2
    if img_dim[1] % 2 == 0:
        trans_y = img_dim[1] / 2
    else:
        trans_y = (img_dim[1] + 1) / 2
    return trans_x, trans_y


def convert_to_of(pos, pos_err, speed, speed_err, height, height_err, focal_len, img_dim):
    """of_library.py:53-75 (reference not executable: eps/trans_x/pos_er undefined).  Expected pin-hole flow
    and its variance, as the formulas there read once the typos are resolved."""
    height = np.asarray(height, np.float64)
    if np.any(height < np.finfo(np.float64).eps):
        raise ValueError(' height over feature is Zero or Negative')
    trans_x, trans_y = pix_trans(img_dim)
    pos = np.asarray(pos, np.float64); pos_err = np.asarray(pos_err, np.float64)
    x_exp = (focal_len - (pos[0, :] - trans_x) / height) * speed[0] / height
    y_exp = (focal_len - (pos[1, :] - trans_y) / height) * speed[1] / height
    x_exp_err = (pos_err[0, :] * speed[0] / height) ** 2 + ((focal_len - pos[0, :] + trans_x) * speed_err[0] / height) ** 2 + \
        ((focal_len - pos[0, :] + trans_x) * speed[0] * height_err / height ** 2) ** 2
    y_exp_err = (pos_err[1, :] * speed[1] / height) ** 2 + ((focal_len - pos[1, :] + trans_y) * speed_err[1] / height) ** 2 + \
        ((focal_len - pos[1, :] + trans_y) * speed[1] * height_err / height ** 2) ** 2
    return [x_exp, y_exp], [x_exp_err, y_exp_err]


def static_immobile(newpos, oldpos, maxspeed, distance, dummy_value):
    """of_library.py:88-92."""
    speed_constraint = (np.abs(newpos - oldpos)) < (maxspeed / distance)
    dummy_constraint = (oldpos) != dummy_value
    stable = speed_constraint * dummy_constraint
    return stable[:, :, 0] * stable[:, :, 1]


def dynamic_immobile(newpos, newpos_err, oldpos, oldpos_err, speed, speed_err, focal_len, dummy_value, height, height_err,
                     img_dim):
    """of_library.py:100-114 (reference not executable: new_pos/new_pos_err undefined)."""
    newpos = np.asarray(newpos, np.float64); oldpos = np.asarray(oldpos, np.float64)
    of_obs = newpos - oldpos
    of_obs_err = np.asarray(oldpos_err, np.float64) ** 2 + np.asarray(newpos_err, np.float64) ** 2
    pos2 = newpos.reshape(-1, 2).T
    err2 = np.broadcast_to(np.asarray(newpos_err, np.float64).reshape(len(pos2[0]), -1), (len(pos2[0]), 2)).T
    of_exp, of_exp_err = convert_to_of(pos2, err2, speed, speed_err, height, height_err, focal_len, img_dim)
    of_exp = np.stack(of_exp, axis=-1).reshape(newpos.shape); of_exp_err = np.stack(of_exp_err, axis=-1).reshape(newpos.shape)
    speed_constraint = ((of_obs - of_exp) ** 2) < (np.broadcast_to(of_obs_err.reshape(len(newpos), 1, -1), newpos.shape) + of_exp_err)
    dummy_constraint = oldpos != dummy_value
    stable = speed_constraint * dummy_constraint
    return stable[:, :, 0] * stable[:, :, 1]


def kmeancluster(points, k):
    """of_library.py:121-136 (cv2.kmeans, 10 iterations / eps 1.0, 10 attempts, random centres).  Lloyd's
    algorithm with the same stopping rule; clustering is outside the hot path (numpy)."""
    pts = np.float32(points).reshape(len(points), -1)
    rng = np.random.default_rng(0)
    best = None
    for _ in range(10):
        centre = pts[rng.choice(len(pts), k, replace=False)].copy()
        for _ in range(10):
            lab = np.argmin(((pts[:, None, :] - centre[None]) ** 2).sum(-1), axis=1)
            new = np.array([pts[lab == i].mean(0) if np.any(lab == i) else centre[i] for i in range(k)])
            shift = np.sqrt(((new - centre) ** 2).sum(-1)).max()
            centre = new
            if shift <= 1.0:
                break
        lab = np.argmin(((pts[:, None, :] - centre[None]) ** 2).sum(-1), axis=1)
        cost = ((pts - centre[lab]) ** 2).sum()
        if best is None or cost < best[0]:
            best = (cost, lab)
    pts_in = np.float32(points)
    return np.array([np.array(pts_in[best[1] == i]) for i in range(k)], dtype=object)


def distancecluster(pointcloud, points, maxdist, clusterlist):
    """of_library.py:146-171 (reference fails at run time: one-argument np.append, `del list[list]`).
    Single-link merge of every cluster that has a member within maxdist (per axis) of the new point."""
    pointcloud = np.asarray(pointcloud, np.float64).reshape(-1, 2)
    clusterlist = [list(np.atleast_1d(c)) for c in clusterlist]
    for i in range(len(points)):
        point = np.asarray(points[i], np.float64).reshape(2)
        near = set(np.where(np.all(np.abs(pointcloud - point) < maxdist, axis=1))[0].tolist()) if len(pointcloud) else set()
        new_index = len(pointcloud)
        fuse = [ci for ci, c in enumerate(clusterlist) if near.intersection(c)]
        merged = [m for ci in fuse for m in clusterlist[ci]] + [new_index]
        clusterlist = [c for ci, c in enumerate(clusterlist) if ci not in fuse] + [merged]
        pointcloud = np.concatenate([pointcloud, point[None]], axis=0)
    return clusterlist, pointcloud


def _disc(mask, x, y, radius, value):
    h, w = mask.shape[:2]
    y0, y1 = max(0, y - radius), min(h, y + radius + 1); x0, x1 = max(0, x - radius), min(w, x + radius + 1)
    if y0 >= y1 or x0 >= x1:
        return
    yy, xx = np.ogrid[y0:y1, x0:x1]
    mask[y0:y1, x0:x1][(yy - y) ** 2 + (xx - x) ** 2 <= radius * radius] = value


def _fill_convex(mask, poly, value):
    poly = np.asarray(poly, np.float64).reshape(-1, 2)
    h, w = mask.shape[:2]
    x0, x1 = int(max(0, np.floor(poly[:, 0].min()))), int(min(w - 1, np.ceil(poly[:, 0].max())))
    y0, y1 = int(max(0, np.floor(poly[:, 1].min()))), int(min(h - 1, np.ceil(poly[:, 1].max())))
    if x0 > x1 or y0 > y1:
        return
    yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1]
    inside_pos = np.ones(yy.shape, bool); inside_neg = np.ones(yy.shape, bool)
    for i in range(len(poly)):
        a, b = poly[i], poly[(i + 1) % len(poly)]
        cr = (b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])
        inside_pos &= cr >= -1e-9; inside_neg &= cr <= 1e-9
    mask[y0:y1 + 1, x0:x1 + 1][inside_pos | inside_neg] = value


def _hull(pts):
    pts = sorted(set(map(tuple, np.asarray(pts, np.float64).reshape(-1, 2))))
    if len(pts) <= 2:
        return np.array(pts)
    def half(seq):
        out = []
        for p in seq:
            while len(out) >= 2 and (out[-1][0] - out[-2][0]) * (p[1] - out[-2][1]) - (out[-1][1] - out[-2][1]) * (p[0] - out[-2][0]) <= 0:
                out.pop()
            out.append(p)
        return out
    lo = half(pts); up = half(pts[::-1])
    return np.array(lo[:-1] + up[:-1])


def boundingboxes(clusterlist, mask, radius):
    """of_library.py:184-192 — zero the rotated minimum-area box of each cluster (single points: a disc)."""
    for cluster in clusterlist:
        cluster = np.asarray(cluster)
        if len(cluster) > 1:
            hull = _hull(cluster)
            best = None
            for i in range(len(hull)):                       # rotating calipers over hull edges
                e = hull[(i + 1) % len(hull)] - hull[i]
                nrm = np.hypot(*e)
                if nrm == 0:
                    continue
                ux = e / nrm; uy = np.array([-ux[1], ux[0]])
                px = hull @ ux; py = hull @ uy
                area = (px.max() - px.min()) * (py.max() - py.min())
                if best is None or area < best[0]:
                    best = (area, ux, uy, px.min(), px.max(), py.min(), py.max())
            if best is None:
                continue
            _, ux, uy, a0, a1, b0, b1 = best
            box = np.array([ux * a0 + uy * b0, ux * a1 + uy * b0, ux * a1 + uy * b1, ux * a0 + uy * b1])
            _fill_convex(mask, np.trunc(box), 0)              # np.int0 truncation (of_library.py:189)
        elif len(cluster) == 1:
            circles(cluster[:, 0, :], mask, radius)


def circles(points, mask, radius):
    """of_library.py:204-208 — zero a disc around each point.  The root file's `y=int(point[0])` (:207) is a
    typo of the older fork's `point[1]` (sensor_precision_experiments/pixhawk_pure_IMU/of_library.py:207);
    the evident intent (x, y) is implemented."""
    for point in points:
        x = int(point[0]); y = int(point[1])
        _disc(mask, x, y, int(radius), 0)


def convexhull(clusterlist, mask, radius):
    """of_library.py:219-226 — zero the convex hull of each cluster (single points: a disc)."""
    for cluster in clusterlist:
        cluster = np.asarray(cluster)
        if len(cluster) > 1:
            _fill_convex(mask, np.array(_hull(cluster), dtype='int32'), 0)
        elif len(cluster) == 1:
            circles(cluster[0, :, :], mask, radius)


def initialize_ft(camera, feature_parameter, lk_parameter, iterations, end_count, vel, vel_err, focal_len, dummy_value,
                  img_dim, weight):
    """of_library.py:231-263 (reference not executable: lk_params/newpos_err/dynamic_immoblie/old_pos_err undefined,
    returns nothing).  `camera` is any iterable of BGR frames (the reference opens cv2.VideoCapture(camera)).
    gray -> goodFeaturesToTrack -> `iterations` x calcOpticalFlowPyrLK, height on the first step, keep immobile
    points, rank with eval_ft.  Returns eval_ft's tuple."""
    if end_count <= 0:
        raise ValueError(' end_count must be a positive number')
    if iterations <= 0:
        raise ValueError(' iterations must be a positive number')
    frames = iter(camera)
    old_gray = cv2.cvtColor(next(frames), cv2.COLOR_BGR2GRAY)
    old_pos = cv2.goodFeaturesToTrack(old_gray, mask=None, **feature_parameter)
    if old_pos is None:
        raise ValueError(' no features found in the first frame')
    old_pos_err = np.zeros((len(old_pos), 1), np.float32)
    vel = np.asarray(vel, np.float64); vel_err = np.asarray(vel_err, np.float64)
    height = height_err = new_pos = new_pos_err = None
    for i in range(iterations):
        frame_gray = cv2.cvtColor(next(frames), cv2.COLOR_BGR2GRAY)
        new_pos, status, new_pos_err = cv2.calcOpticalFlowPyrLK(old_gray, frame_gray, old_pos, None, **lk_parameter)
        if i == 0:
            n = len(new_pos)
            height, height_err = calc_height((new_pos - old_pos).reshape(n, 2), np.broadcast_to(new_pos_err, (n, 2)),
                                             np.broadcast_to(vel, (n, 3)), np.broadcast_to(vel_err, (n, 3)), focal_len,
                                             new_pos.reshape(n, 2), np.broadcast_to(new_pos_err, (n, 2)))
        keep = status.reshape(-1) == 1
        height, height_err = height[keep], height_err[keep]
        old_pos = new_pos[keep].reshape(-1, 1, 2); old_pos_err = new_pos_err[keep]; new_pos_err = new_pos_err[keep]
        new_pos = old_pos
        old_gray = frame_gray
        if len(old_pos) <= end_count:
            break
    return eval_ft(weight, height, height_err, new_pos.reshape(-1, 2), new_pos_err.reshape(-1), img_dim)


def calc_height(of, of_eff, vel, vel_err, focal_len, newpos, newpos_err):
    """of_library.py:270-286 (reference not executable: of_err/new_pos/heigh_y_err undefined).  Height from
    flow with the pin-hole model, mean of the x and y estimates, summed variances."""
    of = np.asarray(of, np.float64); of_err = np.asarray(of_eff, np.float64)
    vel = np.asarray(vel, np.float64); vel_err = np.asarray(vel_err, np.float64)
    new_pos = np.asarray(newpos, np.float64); new_pos_err = np.asarray(newpos_err, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        height_x = (focal_len * vel[:, 0] - new_pos[:, 0] * vel[:, 2]) / of[:, 0]
        height_y = (focal_len * vel[:, 1] - new_pos[:, 1] * vel[:, 2]) / of[:, 1]
        hx_err = (focal_len * vel_err[:, 0] / of[:, 0]) ** 2 + ((focal_len * vel[:, 0] - new_pos[:, 0] * vel[:, 2]) * of_err[:, 0] / of[:, 0] ** 2) ** 2 \
            + (new_pos_err[:, 0] * vel[:, 2] / of[:, 0]) ** 2 + (new_pos[:, 0] * vel_err[:, 2] / of[:, 0]) ** 2
        hy_err = (focal_len * vel_err[:, 1] / of[:, 1]) ** 2 + ((focal_len * vel[:, 1] - new_pos[:, 1] * vel[:, 2]) * of_err[:, 1] / of[:, 1] ** 2) ** 2 \
            + (new_pos_err[:, 1] * vel[:, 2] / of[:, 1]) ** 2 + (new_pos[:, 1] * vel_err[:, 2] / of[:, 1]) ** 2
    return 0.5 * (height_x + height_y), hx_err + hy_err


def eval_ft(weight, height, height_err, new_pos, new_pos_err, img_dim):
    """of_library.py:291-317 (reference not executable: height_err_norm used before assignment, trans undefined).
    Weighted score of normalised height, height variance, centre distance and track error; ascending argsort."""
    height = np.asarray(height, np.float64); height_err = np.asarray(height_err, np.float64)
    new_pos = np.asarray(new_pos, np.float64).reshape(-1, 2); new_pos_err = np.asarray(new_pos_err, np.float64).reshape(-1)
    def norm(a):
        rng_ = np.amax(a) - np.amin(a)
        return (a - np.amin(a)) / rng_ if rng_ > 0 else np.zeros_like(a)
    trans = pix_trans(img_dim)
    height_norm = norm(height); height_err_norm = norm(height_err)
    quad = (new_pos[:, 0] - trans[0]) ** 2 + (new_pos[:, 1] - trans[1]) ** 2
    dist_norm = quad / np.amax(quad) if np.amax(quad) > 0 else np.zeros_like(quad)
    new_pos_err_norm = norm(new_pos_err)
    best_ft = weight[0] * (1 - height_norm) + weight[1] * height_err_norm + weight[2] * (1 - dist_norm) + weight[3] * new_pos_err_norm
    idx = best_ft.argsort()
    return height[idx], height_err[idx], new_pos[idx], new_pos_err[idx]


def read_yaml_imu(yamlfile):
    """of_library.py:327-351 (py2 `file()`, unsafe yaml.load of python-tagged ROS messages).  Reads the same
    logs with a SAFE loader that maps the `!!python/object` tags to plain dicts; entries are
    [t, orientation, orientation_cov, linear_acc, linear_acc_cov, angular_velocity, angular_velocity_cov],
    built from the last message backwards like the reference."""
    from .ingest import load_ros_yaml

    def vec(v, keys):
        return [getattr(v, k) for k in keys]

    stack = []
    for entry in reversed(load_ros_yaml(yamlfile)):
        st = entry.header.stamp
        stack.append([st.secs + float(st.nsecs / 10 ** 6), vec(entry.orientation, "xyzw"),
                      getattr(entry, "orientation_covariance", None), vec(entry.linear_acceleration, "xyz"),
                      getattr(entry, "linear_acceleration_covariance", None), vec(entry.angular_velocity, "xyz"),
                      getattr(entry, "angular_velocity_covariance", None)])
    return stack


def r_tilde(x, u, n, v, dist=None):
    """of_library.py:365-386 — cos of the angle between -p x v and p x (u,0), sign-flipped where p.n < 0, and the
    distance ratio d_i; r = d = 1 where a norm vanishes.  One thread per point (k_feasibility).
    Called with four arguments on homogeneous 3-vectors it is the older fork's form that of_module.py:125 uses
    (sensor_precision_experiments/pixhawk_pure_IMU/of_library.py:365-380)."""
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    if len(x) == 0:
        return np.zeros(0), np.ones(0)
    ctx = ofk.default_context()
    if dist is None:
        if x.shape[1] != 3 or not np.all(x[:, 2] == 1) or (u.shape[1] == 3 and np.any(u[:, 2] != 0)):
            raise ValueError("4-argument r_tilde expects x = [x, y, 1] and u = [ux, uy, 0] rows (of_module.py:96-116)")
        return ctx.feasibility(ofk.FEAS_LEGACY, x[:, :2], u[:, :2], n, v)
    return ctx.feasibility(ofk.FEAS_RTILDE, x[:, :2], u[:, :2], n, v, dist=float(dist))
