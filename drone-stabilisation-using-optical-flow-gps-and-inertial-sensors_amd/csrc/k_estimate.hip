// k_estimate.hip — float64 estimation stages, batched: flow model, feasibility, velocity least squares,
// IMU propagation, lever-arm/rotation, Kalman predict/update, Monte-Carlo error simulation.  gfx950.
//
// These stages move a few KB per problem; they are launch/latency bound, so each kernel handles a whole batch
// (one 256-thread block per problem for the reductions, one thread per problem for the 3-vector algebra).
// The 3N x 3 least-squares problem is solved through its 3x3 normal equations, accumulated in fp64 with
// wavefront shuffles (no MFMA: there is no dense contraction here), diagonalised by cyclic Jacobi:
//   M = sum sA^2 (|p|^2 I - p p^T),  g = sum sA sB X^T X q,   v = sum_k q_k (q_k . g) / lambda_k
// which also yields lstsq's singular values s = sqrt(lambda) and the rank.  The residual sum of squares is
// recomputed in a second pass over the points (the closed form cancels catastrophically when R ~ 0).
#include "ofk_internal.h"

struct Acc { double m00, m01, m02, m11, m12, m22, g0, g1, g2, bb, cnt; };

__device__ __forceinline__ void acc_zero(Acc &a) { a.m00 = a.m01 = a.m02 = a.m11 = a.m12 = a.m22 = a.g0 = a.g1 = a.g2 = a.bb = a.cnt = 0.0; }

// p = (x,y,1); c = p x q  (= [p]x q)
__device__ __forceinline__ void cross_p(double x, double y, double q0, double q1, double q2, double &c0, double &c1, double &c2)
{
    c0 = y * q2 - q1; c1 = q0 - x * q2; c2 = x * q1 - y * q0;
}

__device__ __forceinline__ void acc_point(Acc &a, double x, double y, double q0, double q1, double q2, double sA, double sB)
{
    double c0, c1, c2, t0, t1, t2;
    cross_p(x, y, q0, q1, q2, c0, c1, c2);            // X q
    cross_p(x, y, c0, c1, c2, t0, t1, t2);            // X X q = -X^T X q
    const double pp = x * x + y * y + 1.0, sa2 = sA * sA, sab = sA * sB;
    a.m00 += sa2 * (pp - x * x); a.m01 += sa2 * (-x * y); a.m02 += sa2 * (-x);
    a.m11 += sa2 * (pp - y * y); a.m12 += sa2 * (-y);     a.m22 += sa2 * (pp - 1.0);
    a.g0 -= sab * t0; a.g1 -= sab * t1; a.g2 -= sab * t2;
    a.bb += sB * sB * (c0 * c0 + c1 * c1 + c2 * c2);
    a.cnt += 1.0;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum `v` over the 256 threads of the block; result valid in every thread.  s_red: >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__device__ __forceinline__ double block_minmax(double v, bool want_max, double *s_red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_xor(v, o);
        v = want_max ? fmax(v, u) : fmin(v, u);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double a = want_max ? fmax(s_red[0], s_red[1]) : fmin(s_red[0], s_red[1]);
    const double b = want_max ? fmax(s_red[2], s_red[3]) : fmin(s_red[2], s_red[3]);
    return want_max ? fmax(a, b) : fmin(a, b);
}

// All eleven sums at once: the wave reductions are independent shuffle chains the scheduler interleaves, and ONE LDS exchange
// (two barriers) replaces eleven; the order of the additions is block_sum's, so the results are the same bit for bit.
__device__ void acc_block_sum(Acc &a, double *)
{
    __shared__ double s_part[4][11];
    double v[11] = {a.m00, a.m01, a.m02, a.m11, a.m12, a.m22, a.g0, a.g1, a.g2, a.bb, a.cnt};
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 11; ++k) s_part[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] = (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k]);
    a.m00 = v[0]; a.m01 = v[1]; a.m02 = v[2]; a.m11 = v[3]; a.m12 = v[4]; a.m22 = v[5]; a.g0 = v[6]; a.g1 = v[7]; a.g2 = v[8]; a.bb = v[9]; a.cnt = v[10];
}

// Cyclic Jacobi on a symmetric 3x3; eigenvalues descending in lam[], eigenvectors in columns of V.
__device__ void jacobi3(double A[3][3], double lam[3], double V[3][3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        // converged: the off-diagonal mass is below 1e-20 of the diagonal (its effect on the eigenvalues, off^2 / gap, is far below
        // one ulp); waiting for an exact zero costs sweeps that change nothing — sometimes all 32
        if (off <= 1e-20 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
    int o[3] = {0, 1, 2};
    double d[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j) if (d[o[j]] < d[o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double Vs[3][3];
    for (int k = 0; k < 3; ++k) { lam[k] = d[o[k]]; for (int i = 0; i < 3; ++i) Vs[i][k] = V[i][o[k]]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
}

// Solve from accumulated sums.  out8 = v[3], (residual filled later), rank, s[3].  Returns rank.
__device__ int solve_from_acc(const Acc &a, double *v, double *s3)
{
    double A[3][3] = {{a.m00, a.m01, a.m02}, {a.m01, a.m11, a.m12}, {a.m02, a.m12, a.m22}};
    double lam[3], V[3][3];
    jacobi3(A, lam, V);
    const double rows = 3.0 * a.cnt;
    // lstsq's cut is s_k <= eps*max(M,N)*s_max; on the squared spectrum the resolvable floor is eps*lambda_max,
    // so singular-value ratios below sqrt(eps*rows) count as zero ("rank").
    const double tol = lam[0] * 2.220446049250313e-16 * (rows > 3.0 ? rows : 3.0);
    int rank = 0;
    v[0] = v[1] = v[2] = 0.0;
    for (int k = 0; k < 3; ++k) {
        s3[k] = lam[k] > 0.0 ? sqrt(lam[k]) : 0.0;
        if (lam[k] > tol && lam[k] > 0.0) {
            const double c = (V[0][k] * a.g0 + V[1][k] * a.g1 + V[2][k] * a.g2) / lam[k];
            v[0] += c * V[0][k]; v[1] += c * V[1][k]; v[2] += c * V[2][k];
            ++rank;
        }
    }
    return rank;
}

__device__ __forceinline__ double resid_point(double x, double y, double q0, double q1, double q2, double sA, double sB,
                                              const double *v)
{
    double c0, c1, c2, w0, w1, w2;
    cross_p(x, y, q0, q1, q2, c0, c1, c2);
    cross_p(x, y, v[0], v[1], v[2], w0, w1, w2);
    const double r0 = sA * w0 - sB * c0, r1 = sA * w1 - sB * c1, r2 = sA * w2 - sB * c2;
    return r0 * r0 + r1 * r1 + r2 * r2;
}

// per-point scalings of the three reference systems (see ofk.h)
__device__ __forceinline__ void point_terms(int variant, double x, double y, double ux, double uy, const double *nrm,
                                            const double *om, double d, double wgt, double &q0, double &q1, double &q2,
                                            double &sA, double &sB)
{
    const double ndp = nrm[0] * x + nrm[1] * y + nrm[2];
    if (variant == OFK_SOLVE_OFMODULE) {
        q0 = ux; q1 = uy; q2 = 0.0;
        sA = 1.0 / wgt; sB = sA / ndp;
    } else {
        double c0, c1, c2;
        cross_p(x, y, om[0], om[1], om[2], c0, c1, c2);       // [p]x omega
        q0 = ux + c0; q1 = uy + c1; q2 = c2;
        if (variant == OFK_SOLVE_SIM) { sA = ndp; sB = d; } else { sA = 1.0; sB = d / ndp; }
    }
}

// ------------------------------------------------------------------------------------------------ generic batched solve
__global__ __launch_bounds__(256) void k_solve(int variant, const double *__restrict__ x, const double *__restrict__ u,
                                               const uint8_t *__restrict__ valid, int n, const double *__restrict__ d,
                                               const double *__restrict__ nrm, const double *__restrict__ omega,
                                               const double *__restrict__ t, const double *__restrict__ wgt,
                                               double *__restrict__ out)
{
    __shared__ double s_red[4];
    __shared__ double s_v[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *xb = x + (size_t)b * n * 2, *ub = u + (size_t)b * n * 2;
    const uint8_t *vb = valid ? valid + (size_t)b * n : nullptr;
    const double *wb = wgt ? wgt + (size_t)b * n : nullptr;
    const double nb[3] = {nrm[3 * b], nrm[3 * b + 1], nrm[3 * b + 2]};
    double ob[3] = {0, 0, 0};
    if (omega) { ob[0] = omega[3 * b]; ob[1] = omega[3 * b + 1]; ob[2] = omega[3 * b + 2]; }
    const double db = d ? d[b] : 1.0;
    Acc a; acc_zero(a);
    for (int i = tid; i < n; i += 256) {
        if (vb && !vb[i]) continue;
        double q0, q1, q2, sA, sB;
        point_terms(variant, xb[2 * i], xb[2 * i + 1], ub[2 * i], ub[2 * i + 1], nb, ob, db, wb ? wb[i] : 1.0, q0, q1, q2, sA, sB);
        acc_point(a, xb[2 * i], xb[2 * i + 1], q0, q1, q2, sA, sB);
    }
    acc_block_sum(a, s_red);
    if (tid == 0) {
        double v[3], s3[3];
        const int rank = a.cnt > 0.0 ? solve_from_acc(a, v, s3) : 0;
        if (a.cnt <= 0.0) { v[0] = v[1] = v[2] = 0.0; s3[0] = s3[1] = s3[2] = 0.0; }
        s_v[0] = v[0]; s_v[1] = v[1]; s_v[2] = v[2]; s_v[3] = (double)rank; s_v[4] = s3[0]; s_v[5] = s3[1]; s_v[6] = s3[2];
        s_v[7] = a.cnt;
    }
    __syncthreads();
    const double v[3] = {s_v[0], s_v[1], s_v[2]};
    double r = 0.0;
    for (int i = tid; i < n; i += 256) {
        if (vb && !vb[i]) continue;
        double q0, q1, q2, sA, sB;
        point_terms(variant, xb[2 * i], xb[2 * i + 1], ub[2 * i], ub[2 * i + 1], nb, ob, db, wb ? wb[i] : 1.0, q0, q1, q2, sA, sB);
        r += resid_point(xb[2 * i], xb[2 * i + 1], q0, q1, q2, sA, sB, v);
    }
    r = block_sum(r, s_red);
    if (tid == 0) {
        double *o = out + (size_t)b * OFK_SOLVE_DOUBLES;
        double vx = v[0], vy = v[1], vz = v[2];
        if (t) {                                              // v - omega x t
            const double *tb = t + 3 * b;
            vx -= ob[1] * tb[2] - ob[2] * tb[1]; vy -= ob[2] * tb[0] - ob[0] * tb[2]; vz -= ob[0] * tb[1] - ob[1] * tb[0];
        }
        o[0] = vx; o[1] = vy; o[2] = vz; o[3] = r; o[4] = s_v[3]; o[5] = s_v[4]; o[6] = s_v[5]; o[7] = s_v[6];
    }
}

void ofk_launch_solve(hipStream_t s, int variant, const double *x, const double *u, const uint8_t *valid, int batch,
                      int n, const double *d, const double *nrm, const double *omega, const double *t,
                      const double *wgt, double *out)
{
    hipLaunchKernelGGL(k_solve, dim3(batch), dim3(256), 0, s, variant, x, u, valid, n, d, nrm, omega, t, wgt, out);
}

// ------------------------------------------------------------------------------------------------ r_tilde (device form)
__device__ __forceinline__ void rtilde_point(double x, double y, double ux, double uy, const double *n, const double *v,
                                             double dist, double &r, double &dd)
{
    double vc0, vc1, vc2, uc0, uc1, uc2;
    cross_p(x, y, v[0], v[1], v[2], vc0, vc1, vc2);
    vc0 = -vc0; vc1 = -vc1; vc2 = -vc2;
    cross_p(x, y, ux, uy, 0.0, uc0, uc1, uc2);
    const double vn = sqrt(vc0 * vc0 + vc1 * vc1 + vc2 * vc2), un = sqrt(uc0 * uc0 + uc1 * uc1 + uc2 * uc2);
    if (un * vn == 0.0) { r = 1.0; dd = 1.0; return; }
    const double iun = 1.0 / un;
    r = (vc0 * uc0 + vc1 * uc1 + vc2 * uc2) * iun / vn;
    const double pn = x * n[0] + y * n[1] + n[2];
    if (pn < 0.0) r = -r;
    dd = pn * vn * iun / dist;
}

// ------------------------------------------------------------------------------------------------ frame-pair solve
// One WAVE per pair: x = (next - c) * scaling, u = (next - prev) * scaling for tracked points (node:229-235),
// optional r_tilde filter (node:238-245), solve (node:257), lever arm + rotation (node:258).
// A single wave moves in wherever one wave of the other slice's response kernel or LK retires; the 256-thread workgroup of rounds
// 1-2 needed four wave slots with ~90 VGPRs each on one CU at the same moment and waited for them (13.6 us alone, 186 us mean /
// 0.8 ms worst beside the response kernel, profiles/r02_kernel_stats_slices2.csv).  The sums are formed exactly as that workgroup
// formed them - four "virtual waves" v = 0..3 take the points v*64 + lane + 256 k, each is reduced by the same shuffle butterfly,
// and the four partial sums are added as (s0 + s1) + (s2 + s3) - so the records are the same bit for bit (fp64 addition is
// not associative; k_stream_fuse and the stand-alone k_solve keep that order too).
__device__ __forceinline__ bool pair_point(const float *pp, const float *np_, int i, double cx, double cy, double scaling, int use_feas,
                                           double feas_T, const double *nrm, const double *vp, double d, double &x, double &y, double &ux,
                                           double &uy)
{
    x = ((double)np_[2 * i] - cx) * scaling; y = ((double)np_[2 * i + 1] - cy) * scaling;
    ux = ((double)np_[2 * i] - (double)pp[2 * i]) * scaling; uy = ((double)np_[2 * i + 1] - (double)pp[2 * i + 1]) * scaling;
    if (use_feas) {
        double r, dd;
        rtilde_point(x, y, ux, uy, nrm, vp, d, r, dd);
        if (!(r <= feas_T)) return false;
    }
    return true;
}

__global__ __launch_bounds__(64) void k_pairs_solve(const float *__restrict__ prev_pts, const float *__restrict__ next_pts,
                                                    const uint8_t *__restrict__ status, const int *__restrict__ counts,
                                                    int pts_stride, const double *__restrict__ sensors, int variant,
                                                    int use_feas, double feas_T, const int *__restrict__ cand_count,
                                                    double *__restrict__ records)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const double *sn = sensors + (size_t)b * OFK_SENSOR_DOUBLES;
    const double d = sn[0], nrm[3] = {sn[1], sn[2], sn[3]}, om[3] = {sn[4], sn[5], sn[6]};
    const double scaling = sn[19], cx = sn[20], cy = sn[21], vp[3] = {sn[22], sn[23], sn[24]};
    const int n = counts[b];
    const float *pp = prev_pts + (size_t)b * pts_stride * 2, *np_ = next_pts + (size_t)b * pts_stride * 2;
    const uint8_t *st = status + (size_t)b * pts_stride;
    __shared__ double part[4][13];                               // wave-reduced sums of the four virtual waves (12: the residual)
#pragma unroll 1
    for (int vw = 0; vw < 4; ++vw) {
        Acc a; acc_zero(a);
        double tracked = 0.0;
        for (int i = vw * 64 + lane; i < n; i += 256) {
            if (!st[i]) continue;
            tracked += 1.0;
            double x, y, ux, uy;
            if (!pair_point(pp, np_, i, cx, cy, scaling, use_feas, feas_T, nrm, vp, d, x, y, ux, uy)) continue;
            double q0, q1, q2, sA, sB;
            point_terms(variant, x, y, ux, uy, nrm, om, d, 1.0, q0, q1, q2, sA, sB);
            acc_point(a, x, y, q0, q1, q2, sA, sB);
        }
        const double v[12] = {a.m00, a.m01, a.m02, a.m11, a.m12, a.m22, a.g0, a.g1, a.g2, a.bb, a.cnt, tracked};
#pragma unroll
        for (int k = 0; k < 12; ++k) { const double w_ = wave_sum(v[k]); if (lane == 0) part[vw][k] = w_; }
    }
    __builtin_amdgcn_wave_barrier();                             // one wave: LDS executes its accesses in order
    double t[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) t[k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
    Acc a;
    a.m00 = t[0]; a.m01 = t[1]; a.m02 = t[2]; a.m11 = t[3]; a.m12 = t[4]; a.m22 = t[5]; a.g0 = t[6]; a.g1 = t[7]; a.g2 = t[8]; a.bb = t[9]; a.cnt = t[10];
    const double tracked = t[11];
    double v[3] = {0, 0, 0}, s3[3] = {0, 0, 0};                  // every lane solves the same 3x3 system: no broadcast, no barrier
    const int rank = a.cnt > 0.0 ? solve_from_acc(a, v, s3) : 0;
#pragma unroll 1
    for (int vw = 0; vw < 4; ++vw) {
        double r = 0.0;
        for (int i = vw * 64 + lane; i < n; i += 256) {
            if (!st[i]) continue;
            double x, y, ux, uy;
            if (!pair_point(pp, np_, i, cx, cy, scaling, use_feas, feas_T, nrm, vp, d, x, y, ux, uy)) continue;
            double q0, q1, q2, sA, sB;
            point_terms(variant, x, y, ux, uy, nrm, om, d, 1.0, q0, q1, q2, sA, sB);
            r += resid_point(x, y, q0, q1, q2, sA, sB, v);
        }
        r = wave_sum(r);
        if (lane == 0) part[vw][12] = r;
    }
    __builtin_amdgcn_wave_barrier();
    const double r = (part[0][12] + part[1][12]) + (part[2][12] + part[3][12]);
    if (lane == 0) {
        double *o = records + (size_t)b * OFK_RECORD_DOUBLES;
        const double *R = sn + 7, *off = sn + 16;
        // v_obs - [w]x offset, then rotate (node:258)
        const double e0 = v[0] - (om[1] * off[2] - om[2] * off[1]);
        const double e1 = v[1] - (om[2] * off[0] - om[0] * off[2]);
        const double e2 = v[2] - (om[0] * off[1] - om[1] * off[0]);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = r; o[4] = (double)rank; o[5] = s3[0]; o[6] = s3[1]; o[7] = s3[2];
        o[8] = R[0] * e0 + R[1] * e1 + R[2] * e2; o[9] = R[3] * e0 + R[4] * e1 + R[5] * e2; o[10] = R[6] * e0 + R[7] * e1 + R[8] * e2;
        o[11] = a.cnt; o[12] = (double)n; o[13] = tracked; o[14] = cand_count ? (double)cand_count[b * OFK_CNT_STRIDE] : 0.0; o[15] = 0.0;
    }
}

// The same solve as a 256-thread workgroup per pair (rounds 1-2's kernel): each of the four waves IS one of the virtual waves above, the
// partial sums meet in LDS - bit-identical records, a quarter of the latency.  For small batches, where the chip has room for it:
// 32 pairs of 2000 points (4K, BASELINE configs[4]) take 0.19 ms one wave per pair and 0.05 ms this way.
__global__ __launch_bounds__(256) void k_pairs_solve_wg(const float *__restrict__ prev_pts, const float *__restrict__ next_pts,
                                                     const uint8_t *__restrict__ status, const int *__restrict__ counts,
                                                     int pts_stride, const double *__restrict__ sensors, int variant,
                                                     int use_feas, double feas_T, const int *__restrict__ cand_count,
                                                     double *__restrict__ records)
{
    __shared__ double s_red[4];
    __shared__ double s_v[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *sn = sensors + (size_t)b * OFK_SENSOR_DOUBLES;
    const double d = sn[0], nrm[3] = {sn[1], sn[2], sn[3]}, om[3] = {sn[4], sn[5], sn[6]};
    const double scaling = sn[19], cx = sn[20], cy = sn[21], vp[3] = {sn[22], sn[23], sn[24]};
    const int n = counts[b];
    const float *pp = prev_pts + (size_t)b * pts_stride * 2, *np_ = next_pts + (size_t)b * pts_stride * 2;
    const uint8_t *st = status + (size_t)b * pts_stride;
    Acc a; acc_zero(a);
    double tracked = 0.0;
    for (int i = tid; i < n; i += 256) {
        if (!st[i]) continue;
        tracked += 1.0;
        const double x = ((double)np_[2 * i] - cx) * scaling, y = ((double)np_[2 * i + 1] - cy) * scaling;
        const double ux = ((double)np_[2 * i] - (double)pp[2 * i]) * scaling, uy = ((double)np_[2 * i + 1] - (double)pp[2 * i + 1]) * scaling;
        if (use_feas) {
            double r, dd;
            rtilde_point(x, y, ux, uy, nrm, vp, d, r, dd);
            if (!(r <= feas_T)) continue;
        }
        double q0, q1, q2, sA, sB;
        point_terms(variant, x, y, ux, uy, nrm, om, d, 1.0, q0, q1, q2, sA, sB);
        acc_point(a, x, y, q0, q1, q2, sA, sB);
    }
    acc_block_sum(a, s_red);
    tracked = block_sum(tracked, s_red);
    if (tid == 0) {
        double v[3] = {0, 0, 0}, s3[3] = {0, 0, 0};
        const int rank = a.cnt > 0.0 ? solve_from_acc(a, v, s3) : 0;
        s_v[0] = v[0]; s_v[1] = v[1]; s_v[2] = v[2]; s_v[3] = (double)rank; s_v[4] = s3[0]; s_v[5] = s3[1]; s_v[6] = s3[2];
    }
    __syncthreads();
    const double v[3] = {s_v[0], s_v[1], s_v[2]};
    double r = 0.0;
    for (int i = tid; i < n; i += 256) {
        if (!st[i]) continue;
        const double x = ((double)np_[2 * i] - cx) * scaling, y = ((double)np_[2 * i + 1] - cy) * scaling;
        const double ux = ((double)np_[2 * i] - (double)pp[2 * i]) * scaling, uy = ((double)np_[2 * i + 1] - (double)pp[2 * i + 1]) * scaling;
        if (use_feas) {
            double rr, dd;
            rtilde_point(x, y, ux, uy, nrm, vp, d, rr, dd);
            if (!(rr <= feas_T)) continue;
        }
        double q0, q1, q2, sA, sB;
        point_terms(variant, x, y, ux, uy, nrm, om, d, 1.0, q0, q1, q2, sA, sB);
        r += resid_point(x, y, q0, q1, q2, sA, sB, v);
    }
    r = block_sum(r, s_red);
    if (tid == 0) {
        double *o = records + (size_t)b * OFK_RECORD_DOUBLES;
        const double *R = sn + 7, *off = sn + 16;
        // v_obs - [w]x offset, then rotate (node:258)
        const double e0 = v[0] - (om[1] * off[2] - om[2] * off[1]);
        const double e1 = v[1] - (om[2] * off[0] - om[0] * off[2]);
        const double e2 = v[2] - (om[0] * off[1] - om[1] * off[0]);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = r; o[4] = s_v[3]; o[5] = s_v[4]; o[6] = s_v[5]; o[7] = s_v[6];
        o[8] = R[0] * e0 + R[1] * e1 + R[2] * e2; o[9] = R[3] * e0 + R[4] * e1 + R[5] * e2; o[10] = R[6] * e0 + R[7] * e1 + R[8] * e2;
        o[11] = a.cnt; o[12] = (double)n; o[13] = tracked; o[14] = cand_count ? (double)cand_count[b * OFK_CNT_STRIDE] : 0.0; o[15] = 0.0;
    }
}

void ofk_launch_pairs_solve(hipStream_t s, const float *prev_pts, const float *next_pts, const uint8_t *status,
                            const int *counts, int pts_stride, const double *sensors, int variant, int use_feas,
                            double feas_T, const int *cand_count, double *records, int batch)
{
    // Large batches run beside kernels that rent the whole register file in wave-sized pieces: only a single wave finds room at once
    // (k_pairs_solve).  A small batch leaves the chip half empty: four waves per pair finish sooner (k_pairs_solve_wg).
    if (batch >= 128)
        hipLaunchKernelGGL(k_pairs_solve, dim3(batch), dim3(64), 0, s, prev_pts, next_pts, status, counts, pts_stride, sensors,
                           variant, use_feas, feas_T, cand_count, records);
    else
        hipLaunchKernelGGL(k_pairs_solve_wg, dim3(batch), dim3(256), 0, s, prev_pts, next_pts, status, counts, pts_stride, sensors,
                           variant, use_feas, feas_T, cand_count, records);
}

__global__ void k_records_f32(const double *__restrict__ rec, float *__restrict__ dst, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double *r = rec + (size_t)b * OFK_RECORD_DOUBLES;
    float *o = dst + (size_t)b * 8;
    o[0] = (float)r[0]; o[1] = (float)r[1]; o[2] = (float)r[2]; o[3] = (float)r[3];
    o[4] = (float)r[11]; o[5] = (float)r[7]; o[6] = (float)r[4]; o[7] = (float)r[12];
}

void ofk_launch_records_f32(hipStream_t s, const double *records, float *dst, int batch)
{
    hipLaunchKernelGGL(k_records_f32, dim3((batch + 63) / 64), dim3(64), 0, s, records, dst, batch);
}

// ------------------------------------------------------------------------------------------------ flow model
__global__ void k_flow_model(const double *__restrict__ x, int n, const double *__restrict__ v,
                             const double *__restrict__ omega, const double *__restrict__ d,
                             const double *__restrict__ nrm, const double *__restrict__ t, double *__restrict__ flow)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *om = omega + 3 * b, *nn = nrm + 3 * b;
    double v0 = v[3 * b], v1 = v[3 * b + 1], v2 = v[3 * b + 2];
    if (t) {                                                  // v + omega x t (simulation.py:9)
        const double *tb = t + 3 * b;
        v0 += om[1] * tb[2] - om[2] * tb[1]; v1 += om[2] * tb[0] - om[0] * tb[2]; v2 += om[0] * tb[1] - om[1] * tb[0];
    }
    const double px = x[((size_t)b * n + i) * 2], py = x[((size_t)b * n + i) * 2 + 1];
    const double k = (nn[0] * px + nn[1] * py + nn[2]) / d[b];
    // omega x p
    const double w0 = om[1] - om[2] * py, w1 = om[2] * px - om[0], w2 = om[0] * py - om[1] * px;
    flow[((size_t)b * n + i) * 2] = k * (v0 - v2 * px) + (w0 - w2 * px);
    flow[((size_t)b * n + i) * 2 + 1] = k * (v1 - v2 * py) + (w1 - w2 * py);
}

void ofk_launch_flow_model(hipStream_t s, const double *x, int batch, int n, const double *v, const double *omega,
                           const double *d, const double *nrm, const double *t, double *flow)
{
    hipLaunchKernelGGL(k_flow_model, dim3((n + 255) / 256, batch), dim3(256), 0, s, x, n, v, omega, d, nrm, t, flow);
}

// ------------------------------------------------------------------------------------------------ feasibility
__global__ void k_feasibility(int variant, const double *__restrict__ x, const double *__restrict__ u, int n,
                              const double *__restrict__ nrm, const double *__restrict__ v,
                              const double *__restrict__ dist, const double *__restrict__ omega,
                              const double *__restrict__ t, double *__restrict__ r, double *__restrict__ dd)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t k = (size_t)b * n + i;
    const double px = x[2 * k], py = x[2 * k + 1], ux = u[2 * k], uy = u[2 * k + 1];
    const double *nn = nrm + 3 * b, *vv = v + 3 * b;
    double rr, dv;
    if (variant == OFK_FEAS_RTILDE) {
        rtilde_point(px, py, ux, uy, nn, vv, dist[b], rr, dv);
    } else if (variant == OFK_FEAS_LEGACY) {
        double vc0, vc1, vc2, uc0, uc1, uc2;
        cross_p(px, py, vv[0], vv[1], vv[2], vc0, vc1, vc2);
        vc0 = -vc0; vc1 = -vc1; vc2 = -vc2;
        cross_p(px, py, ux, uy, 0.0, uc0, uc1, uc2);
        const double vn = sqrt(vc0 * vc0 + vc1 * vc1 + vc2 * vc2), iun = 1.0 / sqrt(uc0 * uc0 + uc1 * uc1 + uc2 * uc2);
        rr = (vc0 * uc0 + vc1 * uc1 + vc2 * uc2) * iun / vn;
        const double pn = px * nn[0] + py * nn[1] + nn[2];
        if (pn < 0.0) rr = -rr;
        dv = pn * vn * iun;
    } else {                                                  // simulation.py:108-120
        const double *om = omega + 3 * b, *tb = t + 3 * b;
        const double l0 = vv[0] - (om[1] * tb[2] - om[2] * tb[1]), l1 = vv[1] - (om[2] * tb[0] - om[0] * tb[2]),
                     l2 = vv[2] - (om[0] * tb[1] - om[1] * tb[0]);
        const double w0 = om[1] - om[2] * py, w1 = om[2] * px - om[0], w2 = om[0] * py - om[1] * px;   // omega x p
        double f0, f1, f2, g0, g1, g2;
        cross_p(px, py, l0, l1, l2, f0, f1, f2);
        cross_p(px, py, ux - w0, uy - w1, -w2, g0, g1, g2);
        const double n1 = sqrt(f0 * f0 + f1 * f1 + f2 * f2), n2 = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
        rr = (f0 * g0 + f1 * g1 + f2 * g2) / (n1 * n2);
        dv = n1 / n2 * (px * nn[0] + py * nn[1] + nn[2]);
    }
    r[k] = rr; dd[k] = dv;
}

void ofk_launch_feasibility(hipStream_t s, int variant, const double *x, const double *u, int batch, int n,
                            const double *nrm, const double *v, const double *dist, const double *omega,
                            const double *t, double *r, double *dd)
{
    hipLaunchKernelGGL(k_feasibility, dim3((n + 255) / 256, batch), dim3(256), 0, s, variant, x, u, n, nrm, v, dist, omega, t, r, dd);
}

// ------------------------------------------------------------------------------------------------ IMU propagation
// One message of optical_fusion.call_imu (node:61-89) applied to one stream's state; dv (nullable, 3 doubles) accumulates the
// dead-reckoning increments R (a - 9.81 n) dt (the control input of the resident filter, pipeline.FilterModel.ekf6).
__device__ __forceinline__ void imu_apply(double *s, const double *m, double *dv)
{
    const double secs = m[0], nsecs = m[1], qx = m[2], qy = m[3], qz = m[4], qw = m[5];
    double R[9];
    R[0] = 1.0 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qw * qz); R[2] = 2 * (qw * qy + qx * qz);
    R[3] = 2 * (qx * qy + qw * qz); R[4] = 1.0 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qw * qx);
    R[6] = 2 * (qx * qz - qw * qy); R[7] = 2 * (qw * qx + qy * qz); R[8] = 1.0 - 2 * (qx * qx + qy * qy);
    const double n0 = R[2], n1 = R[5], n2 = R[8];             // R [0,0,1]
    if (s[5] != 0.0) {                                         // first message (node:72-75)
        s[3] = nsecs / 1e9; s[4] = secs; s[5] = 0.0;
    } else {
        const double cur = (secs - s[4]) + nsecs / 1e9;
        const double el = cur - s[3];
        const double a0 = m[12] - 9.81 * n0, a1 = m[13] - 9.81 * n1, a2 = m[14] - 9.81 * n2;
        const double d0 = (R[0] * a0 + R[1] * a1 + R[2] * a2) * el, d1 = (R[3] * a0 + R[4] * a1 + R[5] * a2) * el,
                     d2 = (R[6] * a0 + R[7] * a1 + R[8] * a2) * el;
        s[0] = s[0] + d0; s[1] = s[1] + d1; s[2] = s[2] + d2;
        if (dv) { dv[0] += d0; dv[1] += d1; dv[2] += d2; }
        s[3] = cur;
    }
    for (int k = 0; k < 9; ++k) s[6 + k] = R[k];
    s[15] = n0; s[16] = n1; s[17] = n2;
    s[18] = m[6]; s[19] = m[7]; s[20] = m[8];
    s[21] = m[9]; s[22] = m[10]; s[23] = m[11];
}

__global__ void k_imu(double *__restrict__ state, const double *__restrict__ msg, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    imu_apply(state + (size_t)b * OFK_IMU_STATE, msg + (size_t)b * OFK_IMU_MSG, nullptr);
}

// The IMU messages a stream received since its last frame, applied in order to the RESIDENT state (one thread per stream; the
// arithmetic is a handful of operations per message).  msgs [batch][max_msgs][OFK_IMU_MSG], counts [batch].
__global__ void k_imu_seq(double *__restrict__ state, double *__restrict__ dv, const double *__restrict__ msgs,
                          const int *__restrict__ counts, int max_msgs, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int n = min(max(counts[b], 0), max_msgs);
    for (int k = 0; k < n; ++k) imu_apply(state + (size_t)b * OFK_IMU_STATE, msgs + ((size_t)b * max_msgs + k) * OFK_IMU_MSG, dv + 3 * (size_t)b);
}

void ofk_launch_imu_seq(hipStream_t s, double *state, double *dv, const double *msgs, const int *counts, int max_msgs, int batch)
{
    hipLaunchKernelGGL(k_imu_seq, dim3((batch + 63) / 64), dim3(64), 0, s, state, dv, msgs, counts, max_msgs, batch);
}

void ofk_launch_imu(hipStream_t s, double *state, const double *msg, int batch)
{
    hipLaunchKernelGGL(k_imu, dim3((batch + 63) / 64), dim3(64), 0, s, state, msg, batch);
}

__global__ void k_post_solve(const double *__restrict__ v_obs, const double *__restrict__ rot, const double *__restrict__ ang,
                             const double *__restrict__ offset, int batch, double *__restrict__ v_uav)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double *v = v_obs + 3 * b, *R = rot + 9 * b, *w = ang + 3 * b, *o = offset + 3 * b;
    const double e0 = v[0] - (w[1] * o[2] - w[2] * o[1]), e1 = v[1] - (w[2] * o[0] - w[0] * o[2]), e2 = v[2] - (w[0] * o[1] - w[1] * o[0]);
    v_uav[3 * b] = R[0] * e0 + R[1] * e1 + R[2] * e2;
    v_uav[3 * b + 1] = R[3] * e0 + R[4] * e1 + R[5] * e2;
    v_uav[3 * b + 2] = R[6] * e0 + R[7] * e1 + R[8] * e2;
}

void ofk_launch_post_solve(hipStream_t s, const double *v_obs, const double *rot, const double *ang,
                           const double *offset, int batch, double *v_uav)
{
    hipLaunchKernelGGL(k_post_solve, dim3((batch + 63) / 64), dim3(64), 0, s, v_obs, rot, ang, offset, batch, v_uav);
}

// ------------------------------------------------------------------------------------------------ Kalman filter
#define KF_MAX 6
// cv2.KalmanFilter.predict(control) (of_module.py:122): x = F x (+ B u), P = F P F^T + Q.  Row-major matrices, ns <= 6.
__device__ void kf_predict_dev(int ns, int nc, const double *F, const double *Bm, const double *Q, const double *u, double *x, double (*P)[KF_MAX])
{
    double xn[KF_MAX], T[KF_MAX][KF_MAX];
    for (int i = 0; i < ns; ++i) { double s = 0; for (int j = 0; j < ns; ++j) s += F[i * ns + j] * x[j]; xn[i] = s; }
    if (Bm && u) for (int i = 0; i < ns; ++i) { double s = 0; for (int j = 0; j < nc; ++j) s += Bm[i * nc + j] * u[j]; xn[i] += s; }
    for (int i = 0; i < ns; ++i) x[i] = xn[i];
    for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) { double s = 0; for (int k = 0; k < ns; ++k) s += F[i * ns + k] * P[k][j]; T[i][j] = s; }
    for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) { double s = 0; for (int k = 0; k < ns; ++k) s += T[i][k] * F[j * ns + k]; P[i][j] = s + Q[i * ns + j]; }
}

// cv2.KalmanFilter.correct(z) (of_module.py:152): K = P H^T (H P H^T + R)^-1, x += K (z - H x), P -= K H P.
__device__ void kf_correct_dev(int ns, int nm, const double *H, const double *Rm, const double *z, double *x, double (*P)[KF_MAX])
{
    double HP[KF_MAX][KF_MAX], S[KF_MAX][KF_MAX], K[KF_MAX][KF_MAX], T[KF_MAX][KF_MAX];      // HP: nm x ns, S: nm x nm, K: ns x nm
    for (int i = 0; i < nm; ++i) for (int j = 0; j < ns; ++j) { double s = 0; for (int k = 0; k < ns; ++k) s += H[i * ns + k] * P[k][j]; HP[i][j] = s; }
    for (int i = 0; i < nm; ++i) for (int j = 0; j < nm; ++j) { double s = 0; for (int k = 0; k < ns; ++k) s += HP[i][k] * H[j * ns + k]; S[i][j] = s + Rm[i * nm + j]; }
    // solve S Y = HP (Y: nm x ns) by Gauss-Jordan with partial pivoting; K = Y^T
    for (int c = 0; c < nm; ++c) {
        int piv = c; double best = fabs(S[c][c]);
        for (int r2 = c + 1; r2 < nm; ++r2) if (fabs(S[r2][c]) > best) { best = fabs(S[r2][c]); piv = r2; }
        if (piv != c) {
            for (int j = 0; j < nm; ++j) { const double tmp = S[c][j]; S[c][j] = S[piv][j]; S[piv][j] = tmp; }
            for (int j = 0; j < ns; ++j) { const double tmp = HP[c][j]; HP[c][j] = HP[piv][j]; HP[piv][j] = tmp; }
        }
        const double inv = 1.0 / S[c][c];
        for (int j = 0; j < nm; ++j) S[c][j] *= inv;
        for (int j = 0; j < ns; ++j) HP[c][j] *= inv;
        for (int r2 = 0; r2 < nm; ++r2) if (r2 != c) {
            const double f = S[r2][c];
            if (f != 0.0) {
                for (int j = 0; j < nm; ++j) S[r2][j] -= f * S[c][j];
                for (int j = 0; j < ns; ++j) HP[r2][j] -= f * HP[c][j];
            }
        }
    }
    for (int i = 0; i < ns; ++i) for (int j = 0; j < nm; ++j) K[i][j] = HP[j][i];
    double innov[KF_MAX];
    for (int i = 0; i < nm; ++i) { double s = 0; for (int k = 0; k < ns; ++k) s += H[i * ns + k] * x[k]; innov[i] = z[i] - s; }
    for (int i = 0; i < ns; ++i) { double s = 0; for (int j = 0; j < nm; ++j) s += K[i][j] * innov[j]; x[i] += s; }
    // P = P - K (H P)   (H P recomputed from the prior P)
    for (int i = 0; i < nm; ++i) for (int j = 0; j < ns; ++j) { double s = 0; for (int k = 0; k < ns; ++k) s += H[i * ns + k] * P[k][j]; T[i][j] = s; }
    for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) { double s = 0; for (int k = 0; k < nm; ++k) s += K[i][k] * T[k][j]; P[i][j] -= s; }
}

__global__ void k_kf(int ns, int nm, int nc, const double *__restrict__ F, const double *__restrict__ Bm,
                     const double *__restrict__ H, const double *__restrict__ Q, const double *__restrict__ Rm,
                     double *__restrict__ xs, double *__restrict__ Ps, const double *__restrict__ us,
                     const double *__restrict__ zs, int batch, int do_predict)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double x[KF_MAX], P[KF_MAX][KF_MAX];
    for (int i = 0; i < ns; ++i) { x[i] = xs[(size_t)b * ns + i]; for (int j = 0; j < ns; ++j) P[i][j] = Ps[((size_t)b * ns + i) * ns + j]; }
    if (do_predict) kf_predict_dev(ns, nc, F, Bm, Q, (Bm && us) ? us + (size_t)b * nc : nullptr, x, P);
    if (zs) kf_correct_dev(ns, nm, H, Rm, zs + (size_t)b * nm, x, P);
    for (int i = 0; i < ns; ++i) { xs[(size_t)b * ns + i] = x[i]; for (int j = 0; j < ns; ++j) Ps[((size_t)b * ns + i) * ns + j] = P[i][j]; }
}

// The per-pair filter of a resident batch (BASELINE configs[2]: "batch of independent frame pairs + per-frame EKF update"): one
// thread per pair, predict (no control) + correct with z = z_sign * (v_obs | v_uav) read from the pair's record; state resident.
__global__ void k_kf_records(int ns, int nm, const double *__restrict__ mats, double *__restrict__ xs, double *__restrict__ Ps,
                             const double *__restrict__ records, double z_sign, int z_source, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double *F = mats, *H = mats + 72, *Q = mats + 108, *Rm = mats + 144;
    double x[KF_MAX], P[KF_MAX][KF_MAX];
    for (int i = 0; i < ns; ++i) { x[i] = xs[(size_t)b * ns + i]; for (int j = 0; j < ns; ++j) P[i][j] = Ps[((size_t)b * ns + i) * ns + j]; }
    kf_predict_dev(ns, 0, F, nullptr, Q, nullptr, x, P);
    const double *r = records + (size_t)b * OFK_RECORD_DOUBLES;
    if (r[4] == 3.0) {                                          // a full-rank solve: there is a measurement
        double z[KF_MAX] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 3; ++k) z[k] = z_sign * r[(z_source ? 8 : 0) + k];
        for (int k = 3; k < nm; ++k) z[k] = z[k - 3];
        kf_correct_dev(ns, nm, H, Rm, z, x, P);
    }
    for (int i = 0; i < ns; ++i) { xs[(size_t)b * ns + i] = x[i]; for (int j = 0; j < ns; ++j) Ps[((size_t)b * ns + i) * ns + j] = P[i][j]; }
}

void ofk_launch_kf_records(hipStream_t s, int ns, int nm, const double *mats, double *x, double *P, const double *records, double z_sign,
                           int z_source, int batch)
{
    hipLaunchKernelGGL(k_kf_records, dim3((batch + 63) / 64), dim3(64), 0, s, ns, nm, mats, x, P, records, z_sign, z_source, batch);
}

void ofk_launch_kf(hipStream_t s, int ns, int nm, int nc, const double *F, const double *Bm, const double *H,
                   const double *Q, const double *Rm, double *x, double *P, const double *u, const double *z, int batch,
                   int do_predict)
{
    hipLaunchKernelGGL(k_kf, dim3((batch + 63) / 64), dim3(64), 0, s, ns, nm, nc, F, Bm, H, Q, Rm, x, P, u, z, batch, do_predict);
}

// ------------------------------------------------------------------------------------------------ fused stream step
// One block per video stream, behind LK: everything between calcOpticalFlowPyrLK and the next frame of the reference's loops,
// with the per-stream filter state RESIDENT on the device —
//   of_module.py:96-152   centre; (synthetic rotational flow :113-114); kalman.predict(control) :122; legacy r_tilde with the
//                         predicted velocity :125; keep tracked points with r >= T :129-131 (uint8 status-1 wraps for lost ones); A_i = [p]x / dist_i system :136-146;
//                         kalman.correct(-v_obs) :152; old_pos = new_pos[keep] :166
//   node:229-261          centre + scale; r_tilde with the dead-reckoned velocity :238-245; solve_lgs :257; lever arm + rotation
//                         :258; self.vel = v_uav :261 (the IMU state's velocity, dead-reckoned again by the next IMU messages)
// The keep mask replaces `status` so that k_update_tracks carries exactly the kept points into the next frame.
__device__ __forceinline__ void legacy_point(double x, double y, double ux, double uy, const double *n, const double *v, double &r, double &dd)
{
    double vc0, vc1, vc2, uc0, uc1, uc2;                          // pixhawk_pure_IMU/of_library.py:365-380 (4-arg r_tilde)
    cross_p(x, y, v[0], v[1], v[2], vc0, vc1, vc2);
    vc0 = -vc0; vc1 = -vc1; vc2 = -vc2;
    cross_p(x, y, ux, uy, 0.0, uc0, uc1, uc2);
    const double vn = sqrt(vc0 * vc0 + vc1 * vc1 + vc2 * vc2), iun = 1.0 / sqrt(uc0 * uc0 + uc1 * uc1 + uc2 * uc2);
    r = (vc0 * uc0 + vc1 * uc1 + vc2 * uc2) * iun / vn;
    const double pn = x * n[0] + y * n[1] + n[2];
    if (pn < 0.0) r = -r;
    dd = pn * vn * iun;
}

struct fuse_args {
    const float *prev_pts, *next_pts; uint8_t *status; const int *counts; int pts_stride;
    const double *sensors; double *imu_state, *imu_dv;
    int ns, nm, nc; const double *F, *Bm, *H, *Q, *Rm; double *kf_x, *kf_P;
    ofk_fusion f; int variant, use_feas; double feas_T;
    double *records, *fused;
};

__device__ __forceinline__ bool fuse_point(const fuse_args &g, int i, const float *pp, const float *np_, int st, double cx, double cy,
                                           double scaling, const double *nrm, const double *om, const double *vp, double d, double &x,
                                           double &y, double &ux, double &uy, double &wgt)
{
    const double X = (double)np_[2 * i], Y = (double)np_[2 * i + 1];
    x = (X - cx) * scaling; y = (Y - cy) * scaling;
    if (g.f.flow == OFK_FLOW_ROTATIONAL) {                       // of_module.py:113-114, on the un-centred pixel positions as written there
        ux = X * Y * om[0] + (1.0 + X * X) * om[1] - Y * om[2];
        uy = -(1.0 + Y * Y) * om[0] + X * Y * om[1] + X * om[2];
    } else {
        ux = (X - (double)pp[2 * i]) * scaling; uy = (Y - (double)pp[2 * i + 1]) * scaling;
    }
    wgt = 1.0;
    if (g.f.keep == OFK_KEEP_LEGACY) {
        double r;
        legacy_point(x, y, ux, uy, nrm, vp, r, wgt);
        // of_module.py:129 `feasibility-(status-1)>=T`: status is cv2's uint8 array (:93), so for a lost point status-1 wraps to 255:
        // r - 255 >= T, which a cosine never reaches for any sensible T; for a tracked point it is r - 0 >= T
        return r - (st ? 0.0 : 255.0) >= g.feas_T;
    }
    if (!st) return false;
    if (g.use_feas) {
        double r, dd;
        rtilde_point(x, y, ux, uy, nrm, vp, d, r, dd);
        return r <= g.feas_T;                                    // node:241-245
    }
    return true;
}

__global__ __launch_bounds__(256) void k_stream_fuse(fuse_args g)
{
    __shared__ double s_red[4];
    __shared__ double s_v[8];
    __shared__ double s_pre[12];                                // nrm 0-2, omega 3-5, prior velocity 6-8
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *sn = g.sensors + (size_t)b * OFK_SENSOR_DOUBLES;
    const double d = sn[0], scaling = sn[19], cx = sn[20], cy = sn[21];
    double *ist = g.f.use_imu && g.imu_state ? g.imu_state + (size_t)b * OFK_IMU_STATE : nullptr;
    double kx[KF_MAX], kP[KF_MAX][KF_MAX];                      // thread 0 only
    if (tid == 0) {
        for (int k = 0; k < 3; ++k) { s_pre[k] = ist ? ist[15 + k] : sn[1 + k]; s_pre[3 + k] = ist ? ist[18 + k] : sn[4 + k]; s_pre[6 + k] = ist ? ist[k] : sn[22 + k]; }
        if (g.f.filter) {
            for (int i = 0; i < g.ns; ++i) { kx[i] = g.kf_x[(size_t)b * g.ns + i]; for (int j = 0; j < g.ns; ++j) kP[i][j] = g.kf_P[((size_t)b * g.ns + i) * g.ns + j]; }
            double u[KF_MAX] = {0, 0, 0, 0, 0, 0};
            if (g.f.control == OFK_CONTROL_IMU && g.imu_dv) for (int k = 0; k < 3; ++k) u[k] = g.imu_dv[3 * (size_t)b + k];
            else for (int k = 0; k < 3; ++k) u[k] = sn[25 + k];
            kf_predict_dev(g.ns, g.nc, g.F, g.nc ? g.Bm : nullptr, g.Q, g.nc ? u : nullptr, kx, kP);
            if (g.f.keep == OFK_KEEP_LEGACY) for (int k = 0; k < 3; ++k) s_pre[6 + k] = kx[k];      // v_new = kalman.predict(...) (of_module.py:122,125)
        }
        if (g.imu_dv) for (int k = 0; k < 3; ++k) g.imu_dv[3 * (size_t)b + k] = 0.0;       // "increments since the last step": a step starts a new interval
    }
    __syncthreads();
    const double nrm[3] = {s_pre[0], s_pre[1], s_pre[2]}, om[3] = {s_pre[3], s_pre[4], s_pre[5]}, vp[3] = {s_pre[6], s_pre[7], s_pre[8]};
    const int n = g.counts[b];
    const float *pp = g.prev_pts + (size_t)b * g.pts_stride * 2, *np_ = g.next_pts + (size_t)b * g.pts_stride * 2;
    uint8_t *st = g.status + (size_t)b * g.pts_stride;
    Acc a; acc_zero(a);
    double tracked = 0.0;
    for (int i = tid; i < n; i += 256) {
        const int s0 = st[i];
        tracked += s0 ? 1.0 : 0.0;
        double x, y, ux, uy, wgt;
        const bool keep = fuse_point(g, i, pp, np_, s0, cx, cy, scaling, nrm, om, vp, d, x, y, ux, uy, wgt);
        st[i] = keep ? 1 : 0;
        if (!keep) continue;
        double q0, q1, q2, sA, sB;
        point_terms(g.variant, x, y, ux, uy, nrm, om, d, wgt, q0, q1, q2, sA, sB);
        acc_point(a, x, y, q0, q1, q2, sA, sB);
    }
    acc_block_sum(a, s_red);
    tracked = block_sum(tracked, s_red);
    const bool solved = a.cnt > (double)g.f.min_solve;          // of_module.py:138: more than 3 points; node:256: at least 3
    if (tid == 0) {
        double v[3] = {0, 0, 0}, s3[3] = {0, 0, 0};
        const int rank = solved ? solve_from_acc(a, v, s3) : 0;
        s_v[0] = v[0]; s_v[1] = v[1]; s_v[2] = v[2]; s_v[3] = (double)rank; s_v[4] = s3[0]; s_v[5] = s3[1]; s_v[6] = s3[2];
    }
    __syncthreads();                                            // the keep flags written above are visible to the whole block from here on
    const double v[3] = {s_v[0], s_v[1], s_v[2]};
    double r = 0.0;
    if (solved)
        for (int i = tid; i < n; i += 256) {
            if (!st[i]) continue;
            const double X = (double)np_[2 * i], Y = (double)np_[2 * i + 1];
            const double x = (X - cx) * scaling, y = (Y - cy) * scaling;
            double ux, uy, wgt = 1.0;
            if (g.f.flow == OFK_FLOW_ROTATIONAL) { ux = X * Y * om[0] + (1.0 + X * X) * om[1] - Y * om[2]; uy = -(1.0 + Y * Y) * om[0] + X * Y * om[1] + X * om[2]; }
            else { ux = (X - (double)pp[2 * i]) * scaling; uy = (Y - (double)pp[2 * i + 1]) * scaling; }
            if (g.f.keep == OFK_KEEP_LEGACY) { double rr; legacy_point(x, y, ux, uy, nrm, vp, rr, wgt); }
            double q0, q1, q2, sA, sB;
            point_terms(g.variant, x, y, ux, uy, nrm, om, d, wgt, q0, q1, q2, sA, sB);
            r += resid_point(x, y, q0, q1, q2, sA, sB, v);
        }
    r = block_sum(r, s_red);
    if (tid == 0) {
        double *o = g.records + (size_t)b * OFK_RECORD_DOUBLES;
        const double *R = ist ? ist + 6 : sn + 7, *off = sn + 16;
        const double e0 = v[0] - (om[1] * off[2] - om[2] * off[1]);                 // v_obs - [w]x offset, then rotate (node:258)
        const double e1 = v[1] - (om[2] * off[0] - om[0] * off[2]);
        const double e2 = v[2] - (om[0] * off[1] - om[1] * off[0]);
        const double vu[3] = {R[0] * e0 + R[1] * e1 + R[2] * e2, R[3] * e0 + R[4] * e1 + R[5] * e2, R[6] * e0 + R[7] * e1 + R[8] * e2};
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = r; o[4] = s_v[3]; o[5] = s_v[4]; o[6] = s_v[5]; o[7] = s_v[6];
        o[8] = vu[0]; o[9] = vu[1]; o[10] = vu[2];
        o[11] = a.cnt; o[12] = (double)n; o[13] = tracked; o[14] = 0.0; o[15] = solved ? 1.0 : 0.0;
        double *fu = g.fused + (size_t)b * 8;
        if (g.f.filter) {
            if (solved) {
                double z[KF_MAX] = {0, 0, 0, 0, 0, 0};
                for (int k = 0; k < 3; ++k) z[k] = g.f.z_sign * (g.f.z_source ? vu[k] : v[k]);
                for (int k = 3; k < g.nm; ++k) z[k] = sn[22 + (k - 3)];               // a second velocity measurement (FilterModel.ekf6(gps=True)): the sensors' prior slot
                kf_correct_dev(g.ns, g.nm, g.H, g.Rm, z, kx, kP);
            }
            double tr = 0.0;
            for (int i = 0; i < g.ns; ++i) { g.kf_x[(size_t)b * g.ns + i] = kx[i]; tr += kP[i][i]; for (int j = 0; j < g.ns; ++j) g.kf_P[((size_t)b * g.ns + i) * g.ns + j] = kP[i][j]; }
            for (int k = 0; k < 6; ++k) fu[k] = k < g.ns ? kx[k] : 0.0;
            fu[6] = tr; fu[7] = solved ? 1.0 : 0.0;
        } else {
            for (int k = 0; k < 3; ++k) fu[k] = solved ? vu[k] : (ist ? ist[k] : 0.0);
            fu[3] = fu[4] = fu[5] = fu[6] = 0.0; fu[7] = solved ? 1.0 : 0.0;
        }
        if (g.f.vel_overwrite && solved && ist) { ist[0] = vu[0]; ist[1] = vu[1]; ist[2] = vu[2]; }   // node:261
    }
}

void ofk_launch_stream_fuse(hipStream_t s, const float *prev_pts, const float *next_pts, uint8_t *status, const int *counts, int pts_stride,
                            const double *sensors, double *imu_state, double *imu_dv, int ns, int nm, int nc, const double *kf_mats,
                            double *kf_x, double *kf_P, const ofk_fusion *f, int variant, int use_feas, double feas_T, double *records,
                            double *fused, int batch)
{
    fuse_args g;
    g.prev_pts = prev_pts; g.next_pts = next_pts; g.status = status; g.counts = counts; g.pts_stride = pts_stride; g.sensors = sensors;
    g.imu_state = imu_state; g.imu_dv = imu_dv; g.ns = ns; g.nm = nm; g.nc = nc;
    g.F = kf_mats; g.Bm = kf_mats + 36; g.H = kf_mats + 72; g.Q = kf_mats + 108; g.Rm = kf_mats + 144; g.kf_x = kf_x; g.kf_P = kf_P;
    g.f = *f; g.variant = variant; g.use_feas = use_feas; g.feas_T = feas_T; g.records = records; g.fused = fused;
    hipLaunchKernelGGL(k_stream_fuse, dim3(batch), dim3(256), 0, s, g);
}

// ------------------------------------------------------------------------------------------------ Monte-Carlo error simulation
// One block per trial (simulation.py:36-66).  truth: v[0..2] omega[3..5] height[6] normal[7..9] t[10..12].
// Counter-based noise for the Monte-Carlo sweeps ("Philox/counter-based noise so CPU and GPU draw identical values").
// Element e of trial t of sweep step s under seed (k0, k1) = output (e & 1) of the Box-Muller transform of Philox4x32-10(counter =
// (e >> 1, t, s, 0), key = (k0, k1)): u1 = ((x0 >> 5) 2^26 + (x1 >> 6) + 0.5) 2^-53, u2 likewise from x2, x3 (never 0 or 1),
// r = sqrt(-2 ln u1), normals r cos(2 pi u2), r sin(2 pi u2).  Nothing is stored: a trial's 10 + 4 n normals are regenerated where they
// are used, any rank can produce any trial (the counter holds the GLOBAL trial index), and oracle/estimation_oracle.py restates the
// same function in numpy (Random123's known answers pin the integer part; the f64 transform agrees to the last ulps of libm).
__device__ __forceinline__ void ofk_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
struct ofk_noise { unsigned k0, k1, step, trial; };
__device__ __forceinline__ double ofk_noise_normal(const ofk_noise &g, unsigned e)
{
    unsigned x[4];
    ofk_philox4x32_10(e >> 1, g.trial, g.step, 0u, g.k0, g.k1, x);
    const double u1 = ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6) + 0.5) * 0x1p-53;
    const double u2 = ((double)(x[2] >> 5) * 67108864.0 + (double)(x[3] >> 6) + 0.5) * 0x1p-53;
    const double r = sqrt(-2.0 * log(u1)), th = 6.283185307179586476925 * u2;
    return (e & 1u) ? r * sin(th) : r * cos(th);
}
__global__ void k_noise_normals(ofk_noise g, int count, double *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = ofk_noise_normal(g, (unsigned)e);
}
void ofk_launch_noise_normals(hipStream_t s, unsigned k0, unsigned k1, unsigned step, unsigned trial, int count, double *out)
{
    ofk_noise g = {k0, k1, step, trial};
    hipLaunchKernelGGL(k_noise_normals, dim3((count + 255) / 256), dim3(256), 0, s, g, count, out);
}

// RNG = false: the caller supplies the normals (z, the reference's np.random.normal draws in their order); RNG = true: they come from
// ofk_noise_normal under (seed, step) with the global trial index trial0 + blockIdx.x - the 4096-wide batches of BASELINE configs[4]
// then need no 262 MB noise tensor per step from the host.
template <bool RNG>
__global__ __launch_bounds__(256) void k_of_simulation(const double *__restrict__ truth, const double *__restrict__ sig,
                                                       const double *__restrict__ pos, const double *__restrict__ true_flow,
                                                       int n, const double *__restrict__ z, double *__restrict__ v_obs,
                                                       double *__restrict__ bound, ofk_noise rng)
{
    __shared__ double s_red[4];
    __shared__ double s_v[8];
    const int trial = blockIdx.x, tid = threadIdx.x;
    rng.trial += (unsigned)trial;
    // element e of this trial's row of 10 + 4 n normals
    auto Z = [&](size_t e) -> double { return RNG ? ofk_noise_normal(rng, (unsigned)e) : z[(size_t)trial * (10 + 4 * (size_t)n) + e]; };
    const double zi[7] = {Z(0), Z(1), Z(2), Z(3), Z(4), Z(5), Z(6)};
    const size_t zf = 7, zp = 7 + 2 * (size_t)n;               // first flow / position normal of the row
    const double lv[3] = {truth[0], truth[1], truth[2]}, av[3] = {truth[3], truth[4], truth[5]}, hgt = truth[6];
    const double nv[3] = {truth[7], truth[8], truth[9]}, tr[3] = {truth[10], truth[11], truth[12]};
    const double ang[3] = {av[0] + sig[0] * zi[0], av[1] + sig[0] * zi[1], av[2] + sig[0] * zi[2]};
    const double trn[3] = {tr[0] + sig[1] * zi[3], tr[1] + sig[1] * zi[4], tr[2] + sig[1] * zi[5]};
    const double h_err = hgt + sig[2] * zi[6];
    const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    const double ne[3] = {nv[0] / nn, nv[1] / nn, nv[2] / nn};      // normal noise is discarded (simulation.py:46)
    Acc a; acc_zero(a);
    for (int i = tid; i < n; i += 256) {
        const double x = pos[2 * i] + sig[4] * Z(zp + 2 * i), y = pos[2 * i + 1] + sig[4] * Z(zp + 2 * i + 1);
        const double ux = true_flow[2 * i] + sig[3] * Z(zf + 2 * i), uy = true_flow[2 * i + 1] + sig[3] * Z(zf + 2 * i + 1);
        double q0, q1, q2, sA, sB;
        point_terms(OFK_SOLVE_SIM, x, y, ux, uy, ne, ang, h_err, 1.0, q0, q1, q2, sA, sB);
        acc_point(a, x, y, q0, q1, q2, sA, sB);
    }
    acc_block_sum(a, s_red);
    if (tid == 0) {
        double v[3], s3[3];
        solve_from_acc(a, v, s3);
        s_v[0] = v[0]; s_v[1] = v[1]; s_v[2] = v[2]; s_v[3] = s3[2];      // min singular value
    }
    __syncthreads();
    const double smin = s_v[3];
    double part2 = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double xp0 = pos[2 * i], xp1 = pos[2 * i + 1];
        const double dx0 = sig[4] * Z(zp + 2 * i), dx1 = sig[4] * Z(zp + 2 * i + 1);         // dxp = (pos_err - pos, 0)
        const double dd0 = sig[3] * Z(zf + 2 * i), dd1 = sig[3] * Z(zf + 2 * i + 1);         // ddotx
        // reference computes dxp / ddotx as differences of the perturbed and true values; reproduce that rounding
        const double pe0 = xp0 + dx0, pe1 = xp1 + dx1, fe0 = true_flow[2 * i] + dd0, fe1 = true_flow[2 * i + 1] + dd1;
        const double e0 = pe0 - xp0, e1 = pe1 - xp1, f0 = fe0 - true_flow[2 * i], f1 = fe1 - true_flow[2 * i + 1];
        const double ndx = nv[0] * xp0 + nv[1] * xp1 + nv[2];
        const double v_err = (h_err - hgt) / hgt * ndx + ((ne[0] - nv[0]) * xp0 + (ne[1] - nv[1]) * xp1 + (ne[2] - nv[2])) +
                             (nv[0] * e0 + nv[1] * e1);
        const double da[3] = {ang[0] - av[0], ang[1] - av[1], ang[2] - av[2]};
        // d_err = ddotx + dxp x av + xp x da + dxp
        const double c0 = e1 * av[2], c1 = -e0 * av[2], c2 = e0 * av[1] - e1 * av[0];         // (e0,e1,0) x av
        double g0, g1, g2;
        cross_p(xp0, xp1, da[0], da[1], da[2], g0, g1, g2);
        const double de0 = f0 + c0 + g0 + e0, de1 = f1 + c1 + g1 + e1, de2 = c2 + g2;
        const double w0 = v_err * lv[0] + hgt * de0, w1 = v_err * lv[1] + hgt * de1, w2 = v_err * lv[2] + hgt * de2;
        double k0, k1, k2;
        cross_p(xp0, xp1, w0, w1, w2, k0, k1, k2);
        const double pe = sqrt(k0 * k0 + k1 * k1 + k2 * k2) / smin;
        part2 += pe * pe;
    }
    part2 = block_sum(part2, s_red);
    if (tid == 0) {
        const double v0 = s_v[0] - (ang[1] * trn[2] - ang[2] * trn[1]);
        const double v1 = s_v[1] - (ang[2] * trn[0] - ang[0] * trn[2]);
        const double v2 = s_v[2] - (ang[0] * trn[1] - ang[1] * trn[0]);
        v_obs[3 * trial] = v0; v_obs[3 * trial + 1] = v1; v_obs[3 * trial + 2] = v2;
        const double avn = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]), trnorm = sqrt(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]);
        bound[trial] = sqrt(part2) + avn * sig[1] + sig[0] * trnorm + sig[0] * sig[1];
    }
}

void ofk_launch_of_simulation(hipStream_t s, const double *truth, const double *sig, const double *pos,
                              const double *true_flow, int n, const double *z, int trials, double *v_obs,
                              double *bound)
{
    hipLaunchKernelGGL(k_of_simulation<false>, dim3(trials), dim3(256), 0, s, truth, sig, pos, true_flow, n, z, v_obs, bound, ofk_noise{0u, 0u, 0u, 0u});
}

// the same with the normals generated on the device: trials trial0 .. trial0 + trials - 1 of sweep step `step` under `seed`
void ofk_launch_of_simulation_rng(hipStream_t s, const double *truth, const double *sig, const double *pos, const double *true_flow, int n,
                                  unsigned long long seed, unsigned step, unsigned trial0, int trials, double *v_obs, double *bound)
{
    hipLaunchKernelGGL(k_of_simulation<true>, dim3(trials), dim3(256), 0, s, truth, sig, pos, true_flow, n, (const double *)nullptr, v_obs, bound,
                       ofk_noise{(unsigned)seed, (unsigned)(seed >> 32), step, trial0});
}

// ------------------------------------------------------------------------------------------------ multi-plane sorting statistics
// feas_simulation (simulation.py:70-104), one block per trial: perturbed inputs -> solve (simulation.py:15-30) -> backward
// feasibility with the solved velocity and forward feasibility with the noisy prior velocity (simulation.py:108-120) -> per-point
// residual norms of both velocities in the rows of the system.  truth: v[0..2] omega[3..5] height[6] normal[7..9] t[10..12]
// true_vel[13..15]; sig: ang_vel, translation, height, flow, position, normal, velocity.  z: the reference's draw order
// (omega 3, t 3, height 1, flow 2n, position 2n, velocity 3, orient 1, orient2 1).  per [trial][6][n] in the reference's return
// order: backward_para, backward_dist, forward_para, forward_dist, backward_res, forward_res.
__device__ __forceinline__ void feas_sim_point(double px, double py, double ux, double uy, const double *vv, const double *om,
                                               const double *tb, const double *nn, double &rr, double &dv)
{
    const double l0 = vv[0] - (om[1] * tb[2] - om[2] * tb[1]), l1 = vv[1] - (om[2] * tb[0] - om[0] * tb[2]),
                 l2 = vv[2] - (om[0] * tb[1] - om[1] * tb[0]);
    const double w0 = om[1] - om[2] * py, w1 = om[2] * px - om[0], w2 = om[0] * py - om[1] * px;   // omega x p
    double f0, f1, f2, g0, g1, g2;
    cross_p(px, py, l0, l1, l2, f0, f1, f2);
    cross_p(px, py, ux - w0, uy - w1, -w2, g0, g1, g2);
    const double n1 = sqrt(f0 * f0 + f1 * f1 + f2 * f2), n2 = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    rr = (f0 * g0 + f1 * g1 + f2 * g2) / (n1 * n2);
    dv = n1 / n2 * (px * nn[0] + py * nn[1] + nn[2]);
}

__global__ __launch_bounds__(256) void k_feas_simulation(const double *__restrict__ truth, const double *__restrict__ sig,
                                                         const double *__restrict__ pos, const double *__restrict__ true_flow,
                                                         int n, const double *__restrict__ z, double *__restrict__ per,
                                                         double *__restrict__ v_obs)
{
    __shared__ double s_red[4];
    __shared__ double s_v[4];
    const int trial = blockIdx.x, tid = threadIdx.x;
    const double *zi = z + (size_t)trial * (12 + 4 * (size_t)n);
    const double ang[3] = {truth[3] + sig[0] * zi[0], truth[4] + sig[0] * zi[1], truth[5] + sig[0] * zi[2]};
    const double trn[3] = {truth[10] + sig[1] * zi[3], truth[11] + sig[1] * zi[4], truth[12] + sig[1] * zi[5]};
    const double h_err = truth[6] + sig[2] * zi[6];
    const double *zf = zi + 7, *zp = zi + 7 + 2 * (size_t)n, *zv = zi + 7 + 4 * (size_t)n;
    const double vel[3] = {truth[13] + sig[6] * zv[0], truth[14] + sig[6] * zv[1], truth[15] + sig[6] * zv[2]};
    // orient_err = normal_sig * N(0, normal_sig) (simulation.py:87-88); normal_err = Ry(o2) Rx(o1) normal (simulation.py:89)
    const double o1 = sig[5] * (sig[5] * zv[3]), o2 = sig[5] * (sig[5] * zv[4]);
    const double c1 = cos(o1), s1 = sin(o1), c2 = cos(o2), s2 = sin(o2);
    const double r0 = truth[7], r1 = c1 * truth[8] - s1 * truth[9], r2 = s1 * truth[8] + c1 * truth[9];      // Rx n
    const double ne[3] = {c2 * r0 + s2 * r2, r1, -s2 * r0 + c2 * r2};                                        // Ry (Rx n)
    Acc a; acc_zero(a);
    for (int i = tid; i < n; i += 256) {
        const double x = pos[2 * i] + sig[4] * zp[2 * i], y = pos[2 * i + 1] + sig[4] * zp[2 * i + 1];
        const double ux = true_flow[2 * i] + sig[3] * zf[2 * i], uy = true_flow[2 * i + 1] + sig[3] * zf[2 * i + 1];
        double q0, q1, q2, sA, sB;
        point_terms(OFK_SOLVE_SIM, x, y, ux, uy, ne, ang, h_err, 1.0, q0, q1, q2, sA, sB);
        acc_point(a, x, y, q0, q1, q2, sA, sB);
    }
    acc_block_sum(a, s_red);
    if (tid == 0) {
        double v[3], s3[3];
        solve_from_acc(a, v, s3);
        s_v[0] = v[0] - (ang[1] * trn[2] - ang[2] * trn[1]);      // v - omega x t (simulation.py:28)
        s_v[1] = v[1] - (ang[2] * trn[0] - ang[0] * trn[2]);
        s_v[2] = v[2] - (ang[0] * trn[1] - ang[1] * trn[0]);
        if (v_obs) { v_obs[3 * trial] = s_v[0]; v_obs[3 * trial + 1] = s_v[1]; v_obs[3 * trial + 2] = s_v[2]; }
    }
    __syncthreads();
    const double vo[3] = {s_v[0], s_v[1], s_v[2]};
    double *o = per + (size_t)trial * 6 * n;
    for (int i = tid; i < n; i += 256) {
        const double x = pos[2 * i] + sig[4] * zp[2 * i], y = pos[2 * i + 1] + sig[4] * zp[2 * i + 1];
        const double ux = true_flow[2 * i] + sig[3] * zf[2 * i], uy = true_flow[2 * i + 1] + sig[3] * zf[2 * i + 1];
        double bp, bd, fp, fd;
        feas_sim_point(x, y, ux, uy, vo, ang, trn, ne, bp, bd);
        feas_sim_point(x, y, ux, uy, vel, ang, trn, ne, fp, fd);
        double q0, q1, q2, sA, sB;
        point_terms(OFK_SOLVE_SIM, x, y, ux, uy, ne, ang, 1.0, 1.0, q0, q1, q2, sA, sB);          // rows A_j = [p]x (n.p), b_j = [p]x (u + [p]x w)
        o[i] = bp; o[n + i] = bd; o[2 * (size_t)n + i] = fp; o[3 * (size_t)n + i] = fd;
        o[4 * (size_t)n + i] = sqrt(resid_point(x, y, q0, q1, q2, sA, 1.0, vo));
        o[5 * (size_t)n + i] = sqrt(resid_point(x, y, q0, q1, q2, sA, 1.0, vel));
    }
}

// np.mean(axis = 0) over the trials: one thread per (quantity, point), trials summed in order
__global__ void k_trial_mean(const double *__restrict__ per, int trials, int cols, double *__restrict__ mean)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int t = 0; t < trials; ++t) s += per[(size_t)t * cols + c];
    mean[c] = s / (double)trials;
}

void ofk_launch_feas_simulation(hipStream_t s, const double *truth, const double *sig, const double *pos, const double *true_flow,
                                int n, const double *z, int trials, double *per, double *mean, double *v_obs)
{
    hipLaunchKernelGGL(k_feas_simulation, dim3(trials), dim3(256), 0, s, truth, sig, pos, true_flow, n, z, per, v_obs);
    hipLaunchKernelGGL(k_trial_mean, dim3((6 * n + 255) / 256), dim3(256), 0, s, per, trials, 6 * n, mean);
}

// overlap(data1, data2) (simulation.py:124-136): common bin edges = np.histogram(hstack, bins)[1] = linspace(min, max, bins + 1)
// (min == max: the range is widened by 0.5 on both sides, as numpy does), both histograms over those edges (bins half open,
// the last one closed), sum of the bin-wise minima.  One block; edges in LDS, every element binary-searches them exactly as
// np.searchsorted does, so the counts are numpy's counts.
#define OVL_MAX_BINS 1024
__global__ __launch_bounds__(256) void k_hist_overlap(const double *__restrict__ d1, int n1, const double *__restrict__ d2, int n2,
                                                      int bins, int *__restrict__ out)
{
    __shared__ double s_edge[OVL_MAX_BINS + 1];
    __shared__ int s_h1[OVL_MAX_BINS], s_h2[OVL_MAX_BINS];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    double lo = 1.0 / 0.0, hi = -1.0 / 0.0;
    for (int i = tid; i < n1 + n2; i += 256) { const double v = i < n1 ? d1[i] : d2[i - n1]; lo = fmin(lo, v); hi = fmax(hi, v); }
    lo = block_minmax(lo, false, s_red); hi = block_minmax(hi, true, s_red);
    if (lo == hi) { lo -= 0.5; hi += 0.5; }
    const double step = (hi - lo) / (double)bins;
    for (int i = tid; i <= bins; i += 256) s_edge[i] = i == bins ? hi : (double)i * step + lo;      // np.linspace: arange * step + start, last = stop
    for (int i = tid; i < bins; i += 256) { s_h1[i] = 0; s_h2[i] = 0; }
    __syncthreads();
    for (int i = tid; i < n1 + n2; i += 256) {
        const double v = i < n1 ? d1[i] : d2[i - n1];
        int a = 0, b = bins;                                    // largest k with edge[k] <= v (searchsorted side='right' - 1)
        while (a < b) { const int m = (a + b + 1) >> 1; if (s_edge[m] <= v) a = m; else b = m - 1; }
        const int k = a == bins ? bins - 1 : a;                 // v == last edge falls into the last bin
        if (v >= s_edge[0] && v <= s_edge[bins]) atomicAdd(i < n1 ? &s_h1[k] : &s_h2[k], 1);
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = tid; i < bins; i += 256) acc += (double)min(s_h1[i], s_h2[i]);
    acc = block_sum(acc, s_red);
    if (tid == 0) *out = (int)acc;
}

void ofk_launch_hist_overlap(hipStream_t s, const double *d1, int n1, const double *d2, int n2, int bins, int *out)
{
    hipLaunchKernelGGL(k_hist_overlap, dim3(1), dim3(256), 0, s, d1, n1, d2, n2, bins, out);
}

// ------------------------------------------------------------------------------------------------ per-feature estimators
// of_library.py:270-286 (calc_height), :53-75 + :100-114 (convert_to_of inside dynamic_immobile), :291-317 (eval_ft), batched
// over track sets: one block per set.  The reference's functions carry undefined names; the formulas are the
// ones they spell out, restated in oracle/estimation_oracle.py.  f64 throughout, same operation order as the oracle.

__global__ __launch_bounds__(256) void k_feature_eval(const double *__restrict__ pos, const double *__restrict__ pos_err,
                                                      const double *__restrict__ oldpos, const double *__restrict__ oldpos_err,
                                                      const int *__restrict__ counts, int stride, const double *__restrict__ vel,
                                                      const double *__restrict__ vel_err, double focal, double dummy, double tx,
                                                      double ty, const double *__restrict__ weight, double *__restrict__ height,
                                                      double *__restrict__ height_err, uint8_t *__restrict__ immobile,
                                                      double *__restrict__ score, int *__restrict__ order, int *__restrict__ flags)
{
    __shared__ double s_red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(max(counts[b], 0), stride);
    const size_t o = (size_t)b * stride;
    const double vx = vel[3 * b], vy = vel[3 * b + 1], vz = vel[3 * b + 2];
    const double vex = vel_err[3 * b], vey = vel_err[3 * b + 1], vez = vel_err[3 * b + 2];
    double hmin = INFINITY, hmax = -INFINITY, emin = INFINITY, emax = -INFINITY, pmin = INFINITY, pmax = -INFINITY, qmax = -INFINITY;
    bool bad = false;
    for (int i = tid; i < n; i += 256) {
        const double px = pos[2 * (o + i)], py = pos[2 * (o + i) + 1], e = pos_err[o + i];
        const double ox = oldpos[2 * (o + i)], oy = oldpos[2 * (o + i) + 1], oe = oldpos_err[o + i];
        const double ofx = px - ox, ofy = py - oy;
        // calc_height: (f v_x - x v_z) / u_x, mean of the x and y estimates, summed variances
        const double nx = focal * vx - px * vz, ny = focal * vy - py * vz;
        const double hx = nx / ofx, hy = ny / ofy;
        const double a0 = focal * vex / ofx, a1 = nx * e / (ofx * ofx), a2 = e * vz / ofx, a3 = px * vez / ofx;
        const double b0 = focal * vey / ofy, b1 = ny * e / (ofy * ofy), b2 = e * vz / ofy, b3 = py * vez / ofy;
        const double h = 0.5 * (hx + hy), he = (a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) + (b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3);
        height[o + i] = h; height_err[o + i] = he;
        // dynamic_immobile: (observed - expected flow)^2 < var(observed) + var(expected), expected from convert_to_of
        if (!(h >= 2.220446049250313e-16)) bad = true;          // the reference raises ValueError on a non-positive height
        const double xe = (focal - (px - tx) / h) * vx / h, ye = (focal - (py - ty) / h) * vy / h;
        const double fx = focal - px + tx, fy = focal - py + ty;
        const double c0 = e * vx / h, c1 = fx * vex / h, c2 = fx * vx * he / (h * h);
        const double d0 = e * vy / h, d1 = fy * vey / h, d2 = fy * vy * he / (h * h);
        const double xee = c0 * c0 + c1 * c1 + c2 * c2, yee = d0 * d0 + d1 * d1 + d2 * d2;
        const double obs_err = oe * oe + e * e;
        const bool okx = (ofx - xe) * (ofx - xe) < obs_err + xee, oky = (ofy - ye) * (ofy - ye) < obs_err + yee;
        immobile[o + i] = (uint8_t)(okx && oky && ox != dummy && oy != dummy);
        const double q = (px - tx) * (px - tx) + (py - ty) * (py - ty);
        hmin = fmin(hmin, h); hmax = fmax(hmax, h); emin = fmin(emin, he); emax = fmax(emax, he);
        pmin = fmin(pmin, e); pmax = fmax(pmax, e); qmax = fmax(qmax, q);
    }
    hmin = block_minmax(hmin, false, s_red); hmax = block_minmax(hmax, true, s_red);
    emin = block_minmax(emin, false, s_red); emax = block_minmax(emax, true, s_red);
    pmin = block_minmax(pmin, false, s_red); pmax = block_minmax(pmax, true, s_red);
    qmax = block_minmax(qmax, true, s_red);
    if (bad) atomicOr(flags + 1, 1);
    // eval_ft: weighted score of normalised height, height variance, centre distance and track error
    const double w0 = weight[0], w1 = weight[1], w2 = weight[2], w3 = weight[3];
    for (int i = tid; i < n; i += 256) {
        const double px = pos[2 * (o + i)], py = pos[2 * (o + i) + 1];
        const double hn = hmax - hmin > 0 ? (height[o + i] - hmin) / (hmax - hmin) : 0.0;
        const double en = emax - emin > 0 ? (height_err[o + i] - emin) / (emax - emin) : 0.0;
        const double pn = pmax - pmin > 0 ? (pos_err[o + i] - pmin) / (pmax - pmin) : 0.0;
        const double q = (px - tx) * (px - tx) + (py - ty) * (py - ty);
        const double dn = qmax > 0 ? q / qmax : 0.0;
        score[o + i] = w0 * (1 - hn) + w1 * en + w2 * (1 - dn) + w3 * pn;
    }
    __syncthreads();
    // ascending stable order (NaN last, as numpy sorts): rank by counting
    for (int i = tid; i < n; i += 256) {
        const double si = score[o + i];
        const bool ni = si != si;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double sj = score[o + j];
            const bool nj = sj != sj;
            const bool less = (!nj && ni) || (!nj && !ni && sj < si) || (nj == ni && (ni || sj == si) && j < i);
            rank += less ? 1 : 0;
        }
        order[o + rank] = i;
    }
}

void ofk_launch_feature_eval(hipStream_t s, const double *pos, const double *pos_err, const double *oldpos, const double *oldpos_err,
                             const int *counts, int batch, int stride, const double *vel, const double *vel_err, double focal,
                             double dummy, double tx, double ty, const double *weight, double *height, double *height_err,
                             uint8_t *immobile, double *score, int *order, int *flags)
{
    hipLaunchKernelGGL(k_feature_eval, dim3(batch), dim3(256), 0, s, pos, pos_err, oldpos, oldpos_err, counts, stride, vel, vel_err, focal,
                       dummy, tx, ty, weight, height, height_err, immobile, score, order, flags);
}

// ------------------------------------------------------------------------------------------------ plane-distance statistics
// velocity_measurment_node:249-252 — d_sorted = np.sort(d); d_diff = consecutive differences; the commented line there marks a
// split where a gap reaches the expected distance error (several ground planes in view).  One block per set: bitonic sort in
// LDS (sets are padded with +inf to a power of two <= 4096), gaps, and the number of gaps >= d_exp_err.
#define DSPLIT_MAX 4096
__global__ __launch_bounds__(256) void k_d_split(const double *__restrict__ d, const int *__restrict__ counts, int stride, double d_exp_err,
                                                 double *__restrict__ sorted, double *__restrict__ diff, int *__restrict__ nsplit)
{
    __shared__ double s[DSPLIT_MAX];
    __shared__ int s_cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(max(counts[b], 0), stride);
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = tid; i < m; i += 256) s[i] = i < n ? d[(size_t)b * stride + i] : INFINITY;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const double a = s[i], c = s[p];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { s[i] = c; s[p] = a; }
                }
            }
            __syncthreads();
        }
    int local = 0;
    for (int i = tid; i < n; i += 256) {
        sorted[(size_t)b * stride + i] = s[i];
        if (i + 1 < n) {
            const double g = s[i + 1] - s[i];
            diff[(size_t)b * stride + i] = g;
            local += g >= d_exp_err ? 1 : 0;
        }
    }
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (tid == 0) nsplit[b] = s_cnt;
}

void ofk_launch_d_split(hipStream_t s, const double *d, const int *counts, int batch, int stride, double d_exp_err, double *sorted,
                        double *diff, int *nsplit)
{
    hipLaunchKernelGGL(k_d_split, dim3(batch), dim3(256), 0, s, d, counts, stride, d_exp_err, sorted, diff, nsplit);
}

// ------------------------------------------------------------------------------------------------ sensor association (ingest)
// evaluate_exp.py:77-95 — for every image time the nearest IMU and range samples (np.argmin(np.abs(values - t)): the FIRST
// minimum), then d = range, R from the IMU quaternion, normal = R e_z, omega = angular velocity, written into the pair's
// sensor row (d, normal, omega, rotation; the other fields are left as they are).  One block per image.
__device__ int nearest_sample(const double *__restrict__ ts, int n, double t, double *s_v, int *s_i)
{
    const int tid = threadIdx.x;
    double best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
        const double d = fabs(ts[i] - t);
        if (d < best) { best = d; bi = i; }                     // strict: the earliest of equal distances stays
    }
    s_v[tid] = best; s_i[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const double v = s_v[tid + o];
            const int j = s_i[tid + o];
            if (v < s_v[tid] || (v == s_v[tid] && j < s_i[tid])) { s_v[tid] = v; s_i[tid] = j; }
        }
        __syncthreads();
    }
    const int r = s_i[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_associate(const double *__restrict__ t_img, int n_imu, const double *__restrict__ imu_t,
                                                   const double *__restrict__ imu_q, const double *__restrict__ imu_w, int n_hgt,
                                                   const double *__restrict__ hgt_t, const double *__restrict__ hgt_r,
                                                   double *__restrict__ sensors, int *__restrict__ imu_idx, int *__restrict__ hgt_idx)
{
    __shared__ double s_v[256];
    __shared__ int s_i[256];
    const int b = blockIdx.x;
    const double t = t_img[b];
    const int ii = nearest_sample(imu_t, n_imu, t, s_v, s_i);
    const int hi = nearest_sample(hgt_t, n_hgt, t, s_v, s_i);
    if (threadIdx.x != 0) return;
    imu_idx[b] = ii; hgt_idx[b] = hi;
    const double x = imu_q[4 * ii], y = imu_q[4 * ii + 1], z = imu_q[4 * ii + 2], w = imu_q[4 * ii + 3];
    double *sr = sensors + (size_t)b * OFK_SENSOR_DOUBLES;
    double R[9];
    R[0] = 1.0 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);       R[2] = 2 * (w * y + x * z);
    R[3] = 2 * (x * y + w * z);       R[4] = 1.0 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);       R[7] = 2 * (w * x + y * z);       R[8] = 1.0 - 2 * (x * x + y * y);
    sr[0] = hgt_r[hi];
    sr[1] = R[2]; sr[2] = R[5]; sr[3] = R[8];                   // R (0,0,1)^T
    sr[4] = imu_w[3 * ii]; sr[5] = imu_w[3 * ii + 1]; sr[6] = imu_w[3 * ii + 2];
    for (int k = 0; k < 9; ++k) sr[7 + k] = R[k];
}

void ofk_launch_associate(hipStream_t s, const double *t_img, int n_img, int n_imu, const double *imu_t, const double *imu_q,
                          const double *imu_w, int n_hgt, const double *hgt_t, const double *hgt_r, double *sensors, int *imu_idx,
                          int *hgt_idx)
{
    hipLaunchKernelGGL(k_associate, dim3(n_img), dim3(256), 0, s, t_img, n_imu, imu_t, imu_q, imu_w, n_hgt, hgt_t, hgt_r, sensors,
                       imu_idx, hgt_idx);
}
