// kmpc_waypoints.hip -- batched look-ahead waypoint generation on gfx950.
//
// Replaces, for B vehicles at once, GPSRefTrajectory.get_waypoints of the reference
// (scripts/gps_utils/ref_gps_traj.py:131-142, 172-218): nearest recorded point to (x, y) over the
// whole path, N+1 linearly interpolated waypoints on the arclength grid (target-velocity mode,
// starting ONE step ahead, :175) or on the time grid (starting at the closest point, :191), heading
// wrap-around fix against the current yaw (:204-218) and the end-of-path stop flag (:182-184).
//
// One wavefront per vehicle.  The path (t, X, Y, psi, cumulative distance; ~6.7k samples, 270 KB)
// is shared by every vehicle and stays L2 / Infinity-Cache resident; the nearest-point pass streams
// it with coalesced 8-byte loads (lane l reads samples l, l+64, ...), i.e. 107 KB of L2 reads per
// vehicle and no HBM traffic beyond the first touch -- the kernel is L2-bandwidth bound.
// Arithmetic mirrors numpy exactly (no FMA contraction in the distance and in np.interp's
// slope*(x - xp[j]) + fp[j]) so that indices match bit for bit and values to the last ulp.
#include "kmpc_common.h"

struct WP {
    int M, B, H;         // path samples, vehicles, horizon (H+1 waypoints)
    int use_vtarget;     // 1: arclength grid with per-vehicle v_target, 0: time grid
    double traj_dt;
    const double *t, *X, *Y, *psi, *s;
    const double *pose;  // [B,3] x, y, yaw
    const double *vt;    // [B] or null
    double *ref;         // [B,H+1,3] x, y, psi
    int32_t *stop;       // [B]
    int32_t *closest;    // [B] or null (diagnostic)
};

// np.interp (numpy/core/src/multiarray/compiled_base.c arr_interp) for one query point
DEV double np_interp(double xq, const double *xp, const double *fp, int M)
{
    if (xq > xp[M - 1]) return fp[M - 1];
    if (xq < xp[0]) return fp[0];
    int lo = 0, hi = M;  // largest j with xp[j] <= xq
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= xq) lo = mid; else hi = mid; }
    const int j = lo;
    if (j == M - 1 || xp[j] == xq) return fp[j];
    const double slope = __ddiv_rn(__dsub_rn(fp[j + 1], fp[j]), __dsub_rn(xp[j + 1], xp[j]));
    double r = __dadd_rn(__dmul_rn(slope, __dsub_rn(xq, xp[j])), fp[j]);
    if (r != r) {
        r = __dadd_rn(__dmul_rn(slope, __dsub_rn(xq, xp[j + 1])), fp[j + 1]);
        if (r != r && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}

__global__ __launch_bounds__(64) void kmpc_waypoints_kernel(WP w)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= w.B) return;
    const double x = w.pose[3 * (size_t)b], y = w.pose[3 * (size_t)b + 1], yaw = w.pose[3 * (size_t)b + 2];
    // ---- closest recorded point: argmin (X-x)^2 + (Y-y)^2, first occurrence (np.argmin) ---------
    double best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < w.M; i += 64) {
        const double dx = __dsub_rn(w.X[i], x), dy = __dsub_rn(w.Y[i], y);
        const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        if (d < best) { best = d; bi = i; }
    }
    const double dmin = dpp_min(best);
    const double cand = (best == dmin) ? (double)bi : 1e18;
    int closest = (int)dpp_min(cand);
    // a non-finite pose (NaN or +-inf from the GPS / plant) makes every distance NaN or +inf: no lane records an index.  np.argmin
    // returns 0 then (first NaN / first of the equal minima), and so does this -- never an index outside the path arrays
    closest = (unsigned)closest < (unsigned)w.M ? closest : 0;
    // ---- look-ahead grid ---------------------------------------------------------------------------
    const int k = lane;
    const bool act = k <= w.H;
    const double *grid = w.use_vtarget ? w.s : w.t;
    const double start = grid[closest];
    double q;
    if (w.use_vtarget) q = __dadd_rn(__dmul_rn(__dmul_rn((double)(k + 1), w.traj_dt), w.vt[b]), start);  // x*dt*v + start, x = 1..H+1
    else q = __dadd_rn(__dmul_rn((double)k, w.traj_dt), start);                                            // h*dt + start, h = 0..H
    double xi = 0, yi = 0, pi_ = 0;
    if (act) {
        xi = np_interp(q, grid, w.X, w.M);
        yi = np_interp(q, grid, w.Y, w.M);
        pi_ = np_interp(q, grid, w.psi, w.M);
    }
    // ---- heading wrap-around fix (:204-218) -------------------------------------------------------------
    const double pnext = dpp_mov0<0x130, 0xf>(pi_);  // lane k+1
    const double dd = (k < w.H) ? fabs(__dsub_rn(pnext, pi_)) : 0.0;
    const double dc = act ? fabs(__dsub_rn(pi_, yaw)) : 0.0;
    const bool check1 = dpp_max(dd) < M_PI, check2 = dpp_max(dc) < M_PI;
    if (!(check1 && check2) && act) {
        const double c0 = pi_, c1 = __dadd_rn(pi_, 2.0 * M_PI), c2 = __dsub_rn(pi_, 2.0 * M_PI);
        const double e0 = fabs(__dsub_rn(c0, yaw)), e1 = fabs(__dsub_rn(c1, yaw)), e2 = fabs(__dsub_rn(c2, yaw));
        double bc = c0, be = e0;               // np.argmin: first minimal candidate in [p, p+2pi, p-2pi]
        if (e1 < be) { bc = c1; be = e1; }
        if (e2 < be) { bc = c2; be = e2; }
        pi_ = bc;
    }
    if (act) {
        double *o = w.ref + ((size_t)b * (w.H + 1) + k) * 3;
        o[0] = xi; o[1] = yi; o[2] = pi_;
    }
    if (k == w.H) w.stop[b] = (xi == w.X[w.M - 1] && yi == w.Y[w.M - 1]) ? 1 : 0;  // :182-184
    if (lane == 0 && w.closest) w.closest[b] = closest;
}

hipError_t kmpc_launch_waypoints(const WP &w, hipStream_t st)
{
    hipLaunchKernelGGL(kmpc_waypoints_kernel, dim3(w.B), dim3(64), 0, st, w);
    return hipGetLastError();
}
