// Per-body hydrodynamic wrench arithmetic for the gfx950 kernels.
//
// One call = one rigid body, one physics step, evaluated in FLOAT64 from the fp32 inputs and rounded to fp32 once, at
// the end - the arithmetic type of the reference's Numba path (numba_hydrodynamics_wrapper.py:40-45 casts every
// input to float64).  Why fp64 on a GPU whose fp64 vector rate is half its fp32 rate:
//   * the wrench is a SUM of terms that cancel - buoyancy against drag along z, the buoyancy-arm, drag-arm and
//     lift-arm torques against the angular drag and the added mass.  Terms evaluated in fp32 are good to 1-2.5e-7 of
//     THEMSELVES; a 70-350x cancellation (seen for one body in ~1e7) then misses the 1e-5 parity gate, and no
//     rearrangement of fp32 arithmetic removes that.  Round 1 of this repository did everything that can be done
//     in fp32 (fp64 islands for the worst cancellations, structural-zero forms, exact identities in |q|^2 - 1) and
//     still had 48 of 692 M margin-gated evaluations above 1e-5.
//   * the kernels are HBM-bound (122-136 B per body against ~600 VALU instructions): the MI355X has 157 TFLOP/s of
//     fp32 and 79 TFLOP/s of fp64 VALU behind 8 TB/s.  Measured: the all-fp64 body costs about
//     what the fp32 + fp64-island body it replaces cost at 1 M bodies, and the design tried in between - an fp32
//     pass plus an fp64 re-evaluation of the rare ill-conditioned bodies - costs MORE: the flagged wavefronts are the
//     tail of every launch (+1.4 us on a 2.7 us launch of 4 096 bodies).
// No MFMA: the path is elementwise per body.
//
// The model (what must come out) is the reference's
//   numba_hydrodynamics.py:9-314      (A1-A11: rotation, submersion + CoB, CoP + projected area, hybrid drag,
//                                      lift, added mass)
//   hydrodynamics_behavior.py:196-226 (A13-A15: finite-difference acceleration, lever-arm torques, sum,
//                                      500 m/s^2 clamp)
// The evaluation is restructured for the GPU (checked against the fp64 oracle: tests/test_numerics_host.py,
// tests/test_parity_gpu.py):
//   * keypoint heights come from row 2 of R only:
//       z_ijk = p_z + i*e_x + j*e_y + k*e_z,  e_a = h_a * R[2][a],  i,j,k in {-1,0,1}
//     so z_min/max = p_z -/+ (|e_x|+|e_y|+|e_z|) and the 27 world points are never formed
//     (numba_hydrodynamics.py:271 builds them with a 3x27 GEMM); the 27 "below the surface" tests are sign bits
//     funnel-shifted into one mask register, the wet count and the lattice-index sums are popcounts of it;
//   * CoB and CoP are kept as BODY-RELATIVE lever arms (cob - p, cop - p): the wrench only ever uses those
//     differences (hydrodynamics_behavior.py:212-214), so p_x and p_y are never needed (nor loaded);
//   * at most one face per axis opposes the flow, selected by sign(R^T v_hat);
//   * sin(2*asin(d)) = 2 d sqrt((1-d)(1+d));
//   * reciprocals and square roots are the fp32 hardware seeds + one Newton step in fp64 (1.4e-14).
// N1 completion (speed <= 1e-6 -> area 0, CoP = CoB) as in oracle/hydro_oracle.py.
//
// This header is compiled for the device by hipcc and, for the CPU-side numerics study only (tests/host_emul), by the
// host compiler; the product never calls the host instantiation.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HYDRO_FN __host__ __device__ __forceinline__
#else
#define HYDRO_FN static inline __attribute__((always_inline))
#endif

namespace hydro {

// thresholds of the model, as the reference's float64 literals
constexpr double kSpeedEps = 1e-6;      // numba_hydrodynamics.py:118,156,170,192,286
constexpr double kLowSpeed = 0.2;       // :154
constexpr double kDryEps = 1e-9;        // :192,225,277
constexpr double kAreaEps = 1e-6;       // :140
constexpr double kHeightEps = 1e-6;     // :92
constexpr double kAxisEps = 1e-6;       // :210
constexpr double kMaxAccel = 500.0;     // hydrodynamics_behavior.py:221
constexpr double kClampEps = 1e-6;      // hydrodynamics_behavior.py:224

// fp64 reciprocal and square root: the fp32 hardware seeds (v_rcp_f32 / v_rsq_f32, 1 ulp = 2^-23) and ONE Newton step
// in fp64, which squares the error: 1.4e-14 relative - five orders of magnitude below the fp32 rounding of the results,
// so even a 1e5-fold cancellation downstream stays at 1e-9.  5 and 7 instructions where the IEEE-exact sequences the
// compiler expands `/` and sqrt() to take ~15 and ~25.  Arguments are within fp32 range wherever the result is used
// (guarded by the model's own 1e-6 thresholds); sqrt64 returns 0 below 1e-30.  Plain libm on the host instantiation.
HYDRO_FN double rcp64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = (double)__builtin_amdgcn_rcpf((float)x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#else
    return 1.0 / x;
#endif
}
HYDRO_FN double sqrt64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = (double)__builtin_amdgcn_rsqf((float)x);
    const double y = x * r;
    return x > 1e-30 ? __builtin_fma(__builtin_fma(-y, y, x), 0.5 * r, y) : 0.0;
#else
    return x > 1e-30 ? sqrt(x) : 0.0;
#endif
}

// single-instruction fp32 forms on the device (v_sqrt_f32 / v_rcp_f32, 1 ulp); plain libm on the host instantiation
HYDRO_FN float fast_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
HYDRO_FN float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

HYDRO_FN uint32_t high_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, sizeof u); return (uint32_t)(u >> 32); }
// mask = (mask << 1) | signbit(z): one v_alignbit_b32 on the device (the sign of a double is bit 31 of its high dword)
HYDRO_FN uint32_t shift_in_sign(uint32_t mask, double z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(mask, high_bits(z), 31);
#else
    return (mask << 1) | (high_bits(z) >> 31);
#endif
}
// bits of the 27-bit keypoint mask (point p = 9 i + 3 j + k sits at bit 26 - p) whose lattice index
// along `axis` (0: i, 1: j, 2: k) equals `value` (0, 1, 2 for -1, 0, +1)
constexpr uint32_t lattice_mask(int axis, int value)
{
    uint32_t m = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const int idx = axis == 0 ? i : (axis == 1 ? j : k);
                if (idx == value) m |= 1u << (26 - (9 * i + 3 * j + k));
            }
    return m;
}

// the bit of lattice point (i,j,k), i,j,k in {-1,0,1}
constexpr uint32_t lattice_bit(int i, int j, int k) { return 1u << (26 - (9 * (i + 1) + 3 * (j + 1) + (k + 1))); }

struct BodyIn {
    float px, py, pz;               // p_x, p_y are never read by the wrench (it does not depend on them)
    float qx, qy, qz, qw;           // unit quaternion xyzw, NOT normalised (N7)
    float vx, vy, vz;
    float wx, wy, wz;
    float dimx, dimy, dimz;
    float cd_lin, cd_ang, damp_lin, damp_ang, lift, am_lin, am_ang;
};

// Everything the reference's solve_hydrodynamics returns (numba_hydrodynamics.py:314), in fp64, with the two centres
// as lever arms.
struct Body {
    double ratio;                   // submersion ratio
    double buoy_z;                  // buoyancy force is (0,0,buoy_z)
    double drag_fx, drag_fy, drag_fz;
    double lift_fx, lift_fy, lift_fz;
    double drag_tx, drag_ty, drag_tz;
    double am_fx, am_fy, am_fz;
    double am_tx, am_ty, am_tz;
    double armb_x, armb_y, armb_z;  // cob - p
    double armp_x, armp_y, armp_z;  // cop - p
    double lin_k, ang_k;            // drag_force = lin_k * v, drag_torque = ang_k * w   (both <= 0)
    bool wet;                       // ratio > 1e-9; the zeros of a dry body (A4) are applied by assemble_wrench / round_components
};

// A1-A11 for one body.  (ax..bz) = linear / angular acceleration.  rho, g: scene scalars
// (hydrodynamics_config.json:2-5 "globals"), doubles as the reference passes Python floats.
// `warp` (uniform over a launch) selects the semantics of the reference's Warp twin where it differs from the Numba
// path (include/hydro.h HYDRO_SEM_WARP - PARITY UNPINNED for that mode); the default is Numba.
//
// `late(anchor, ...)` supplies what only the LAST block (A10, added mass) consumes: the six accelerations and the two
// added-mass coefficients.  It is called right where they are needed, with a value computed in the middle of the body
// (`anchor`): a kernel may tie its loads of those inputs to it so that they are issued late and their registers are
// not live through the first two thirds of the arithmetic (hydro_kernels.hip, LATE template argument).  The results
// do not depend on where the inputs come from.
template <class Late>
HYDRO_FN Body solve_body_with(const BodyIn& b, Late&& late, double rho, double g, bool warp)
{
    Body o;
    // ---- A1: rotation matrix (numba_hydrodynamics.py:14-49), the quaternion used as given (N7) ----
    const double qx = b.qx, qy = b.qy, qz = b.qz, qw = b.qw;
    const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const double xx = qx * x2, yy = qy * y2, zz = qz * z2;
    const double sx = qw * x2, sy = qw * y2, sz = qw * z2;
    const double r00 = 1.0 - (yy + zz), r01 = __builtin_fma(qx, y2, -sz), r02 = __builtin_fma(qx, z2, sy);
    const double r10 = __builtin_fma(qx, y2, sz), r11 = 1.0 - (xx + zz), r12 = __builtin_fma(qy, z2, -sx);
    const double r20 = __builtin_fma(qx, z2, -sy), r21 = __builtin_fma(qy, z2, sx), r22 = 1.0 - (xx + yy);
    const double dx = b.dimx, dy = b.dimy, dz = b.dimz;
    const double hx = 0.5 * dx, hy = 0.5 * dy, hz = 0.5 * dz;
    const double axy_ = dx * dy, vol = axy_ * dz;          // face area of the z faces, volume

    // ---- A3: vertical extent, submersion ratio (:86-96) ----
    const double ex = hx * r20, ey = hy * r21, ez = hz * r22;
    const double extent = fabs(ex) + fabs(ey) + fabs(ez);
    const double pz = (double)b.pz + 0.0;               // -0.0 -> +0.0: no keypoint height below can then be -0.0
    const double zlo = pz - extent, zhi = pz + extent;  // lowest / highest keypoint
    const bool dry = zlo >= 0.0, full = zhi <= 0.0;
    const double height = zhi - zlo;
    double ratio = fmin(1.0, -zlo * rcp64(height));
    if (height < kHeightEps) ratio = 1.0;               // z_lo < 0 is known here
    if (full) ratio = 1.0;
    if (dry) ratio = 0.0;
    o.ratio = ratio;
    o.wet = ratio > kDryEps;                            // A4 (:277-279)

    // ---- A3: centre of buoyancy from integer lattice sums (:69-84,99-103) ----
    // z_ijk < 0 for lattice index (i,j,k); S_a = sum of index a over the wet points.  The 27 sign bits are
    // funnel-shifted into ONE mask register (one v_alignbit_b32 per keypoint, on the high dword of the double); the
    // count and the three index sums are popcounts of that mask against compile-time masks.
    const double zi[3] = {pz - ex, pz, pz + ex};
    uint32_t wetmask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double zij[3] = {zi[i] - ey, zi[i], zi[i] + ey};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wetmask = shift_in_sign(wetmask, zij[j] - ez);
            wetmask = shift_in_sign(wetmask, zij[j]);
            wetmask = shift_in_sign(wetmask, zij[j] + ez);
        }
    }
    const int cnt = __builtin_popcount(wetmask);
    const int s_i = __builtin_popcount(wetmask & lattice_mask(0, 2)) - __builtin_popcount(wetmask & lattice_mask(0, 0));
    const int s_j = __builtin_popcount(wetmask & lattice_mask(1, 2)) - __builtin_popcount(wetmask & lattice_mask(1, 0));
    const int s_k = __builtin_popcount(wetmask & lattice_mask(2, 2)) - __builtin_popcount(wetmask & lattice_mask(2, 0));
    const double inv_cnt = (!dry && !full && cnt > 0) ? rcp64((double)cnt) : 0.0;    // fully in / dry: cob = position
    const double lbx = hx * ((double)s_i * inv_cnt), lby = hy * ((double)s_j * inv_cnt), lbz = hz * ((double)s_k * inv_cnt);
    o.armb_x = r00 * lbx + r01 * lby + r02 * lbz;
    o.armb_y = r10 * lbx + r11 * lby + r12 * lbz;
    o.armb_z = r20 * lbx + r21 * lby + r22 * lbz;

    // ---- A5: buoyancy (:282) ----
    o.buoy_z = rho * (ratio * vol) * g;

    // ---- A6: speed and direction (:285-289) ----
    const double vx = b.vx, vy = b.vy, vz = b.vz;
    const double speed = sqrt64(vx * vx + vy * vy + vz * vz);
    const bool moving = speed > kSpeedEps;
    const double inv_speed = moving ? rcp64(speed) : 0.0;
    const double nx = vx * inv_speed, ny = vy * inv_speed, nz = vz * inv_speed;          // v_hat (0 at rest)

    // ---- A7: projected area + centre of pressure (:108-143) ----
    // u = R^T v_hat; face (axis a, sign s) has alignment -s*u_a and centre height p_z + s*e_a: the face of axis a that
    // opposes the flow has s_a = -sign(u_a).  At rest u = 0 and nothing counts (N1 completion: area 0, cop = cob).
    const double ux = r00 * nx + r10 * ny + r20 * nz;
    const double uy = r01 * nx + r11 * ny + r21 * nz;
    const double uz = r02 * nx + r12 * ny + r22 * nz;
    const double fsx = (ux < 0.0) ? 1.0 : -1.0, fsy = (uy < 0.0) ? 1.0 : -1.0, fsz = (uz < 0.0) ? 1.0 : -1.0;
    // (the six face centres ARE lattice points - (+-1,0,0), (0,+-1,0), (0,0,+-1) - so "centre below the surface" is a
    // bit of the keypoint mask: the same value p_z +- e_a, the same sign bit)
    const bool wfx = wetmask & ((ux < 0.0) ? lattice_bit(1, 0, 0) : lattice_bit(-1, 0, 0));
    const bool wfy = wetmask & ((uy < 0.0) ? lattice_bit(0, 1, 0) : lattice_bit(0, -1, 0));
    const bool wfz = wetmask & ((uz < 0.0) ? lattice_bit(0, 0, 1) : lattice_bit(0, 0, -1));
    const double fax = ((ux != 0.0) && wfx) ? fabs(ux) * (dy * dz) : 0.0;
    const double fay = ((uy != 0.0) && wfy) ? fabs(uy) * (dx * dz) : 0.0;
    const double faz = ((uz != 0.0) && wfz) ? fabs(uz) * axy_ : 0.0;
    const double area = fax + fay + faz;
    const bool has_area = area > kAreaEps;
    const double inv_area = has_area ? rcp64(area) : 0.0;
    // body-frame CoP arm: sum of (face centre x its share of the area); without area cop = cob (:115,140).  The
    // choice is made on the body-frame vector, so one rotation serves both cases.
    const double lpx = has_area ? fsx * hx * (fax * inv_area) : lbx;
    const double lpy = has_area ? fsy * hy * (fay * inv_area) : lby;
    const double lpz = has_area ? fsz * hz * (faz * inv_area) : lbz;
    o.armp_x = r00 * lpx + r01 * lpy + r02 * lpz;
    o.armp_y = r10 * lpx + r11 * lpy + r12 * lpz;
    o.armp_z = r20 * lpx + r21 * lpy + r22 * lpz;

    // ---- A8: hybrid drag (:146-182).  -(1/2 rho s^2 Cd A) v_hat = -(1/2 rho s Cd A) v, so both parts scale v ----
    const double half_rho = 0.5 * rho;
    const double lin_quad = moving ? half_rho * speed * ((double)b.cd_lin * area) : 0.0;
    const double lin_scale = (speed < kLowSpeed) ? speed * 5.0 : 1.0;                     // min(1, s / 0.2)
    o.lin_k = -(lin_quad + (double)b.damp_lin * lin_scale) * ratio;
    o.drag_fx = o.lin_k * vx; o.drag_fy = o.lin_k * vy; o.drag_fz = o.lin_k * vz;
    const double ox = b.wx, oy = b.wy, oz = b.wz;
    const double wspeed = sqrt64(ox * ox + oy * oy + oz * oz);
    const double ang_quad = (wspeed > kSpeedEps) ? half_rho * wspeed * ((double)b.cd_ang * vol) : 0.0;   // note: volume, not area
    const double ang_scale = (wspeed < kLowSpeed) ? wspeed * 5.0 : 1.0;
    o.ang_k = -(ang_quad + (double)b.damp_ang * ang_scale) * ratio;
    o.drag_tx = o.ang_k * ox; o.drag_ty = o.ang_k * oy; o.drag_tz = o.ang_k * oz;

    // ---- A9: lift (:185-217) ----
    // up = R[:,2]; d = clamp(-up.v_hat) = clamp(-u_z); C_L = sin(2 asin d) = 2 d sqrt((1-d)(1+d));
    // dir = (axis / |axis|) x v_hat with axis = v_hat x up; nothing if speed < 1e-6 or |axis| < 1e-6.
    {
        const double d = fmin(1.0, fmax(-1.0, -uz));
        const double axx = ny * r22 - nz * r12, axy = nz * r02 - nx * r22, axz = nx * r12 - ny * r02;
        const double n2 = axx * axx + axy * axy + axz * axz;
        const bool lift_on = !(speed < kSpeedEps) && !(n2 < kAxisEps * kAxisEps);
        // C_L / |axis| from ONE reciprocal and ONE square root:  2 d sqrt((1-d)(1+d) / |axis|^2)
        const double cl_over_n = lift_on ? 2.0 * d * sqrt64(fmax(0.0, (1.0 - d) * (1.0 + d)) * rcp64(n2)) : 0.0;
        const double k = (half_rho * (speed * speed) * (area * (double)b.lift)) * (cl_over_n * ratio);
        // axis x v_hat = up |v_hat|^2 - v_hat (v_hat . up) = up - u_z v_hat      (|v_hat|^2 = 1 to 1e-16)
        o.lift_fx = k * __builtin_fma(-uz, nx, r02); o.lift_fy = k * __builtin_fma(-uz, ny, r12); o.lift_fz = k * __builtin_fma(-uz, nz, r22);
    }

    // ---- A10: added mass (:220-253; diagonal of numba_hydrodynamics_wrapper.py:101-112) ----
    {
        double ax, ay, az, bx, by, bz;                                          // world-frame accelerations
        float am_lin, am_ang;
        late(area, ax, ay, az, bx, by, bz, am_lin, am_ang);
        double alx, aly, alz, blx, bly, blz;                                    // accelerations in the "local" frame
        if (warp) {                                                             // N3: quat_rotate = R (warp_hydrodynamics.py:216-217)
            alx = r00 * ax + r01 * ay + r02 * az; aly = r10 * ax + r11 * ay + r12 * az; alz = r20 * ax + r21 * ay + r22 * az;
            blx = r00 * bx + r01 * by + r02 * bz; bly = r10 * bx + r11 * by + r12 * bz; blz = r20 * bx + r21 * by + r22 * bz;
        } else {                                                                // R^T (numba_hydrodynamics.py:229-230)
            alx = r00 * ax + r10 * ay + r20 * az; aly = r01 * ax + r11 * ay + r21 * az; alz = r02 * ax + r12 * ay + r22 * az;
            blx = r00 * bx + r10 * by + r20 * bz; bly = r01 * bx + r11 * by + r21 * bz; blz = r02 * bx + r12 * by + r22 * bz;
        }
        const double rv = vol * rho;
        const double kf = -(rv * (double)am_lin) * ratio, kt = -(rv * (double)am_ang) * ratio;
        const double glx = kf * alx, gly = kf * aly, glz = kf * alz;
        const double tlx = kt * (dy * dy + dz * dz) * blx, tly = kt * (dx * dx + dz * dz) * bly, tlz = kt * (dx * dx + dy * dy) * blz;
        o.am_fx = r00 * glx + r01 * gly + r02 * glz; o.am_fy = r10 * glx + r11 * gly + r12 * glz; o.am_fz = r20 * glx + r21 * gly + r22 * glz;
        o.am_tx = r00 * tlx + r01 * tly + r02 * tlz; o.am_ty = r10 * tlx + r11 * tly + r12 * tlz; o.am_tz = r20 * tlx + r21 * tly + r22 * tlz;
    }
    return o;
}

// A1-A11 with the accelerations and the added-mass coefficients at hand (component mode, host instantiation).
HYDRO_FN Body solve_body(const BodyIn& b, double ax, double ay, double az, double bx, double by, double bz,
                         double rho, double g, bool warp = false)
{
    return solve_body_with(b, [&](double, double& oax, double& oay, double& oaz, double& obx, double& oby, double& obz,
                                  float& am_lin, float& am_ang) {
        oax = ax; oay = ay; oaz = az; obx = bx; oby = by; obz = bz; am_lin = b.am_lin; am_ang = b.am_ang;
    }, rho, g, warp);
}

struct Wrench {
    float fx, fy, fz, tx, ty, tz;
    float k_lin, k_ang;             // clamped drag coefficients for the implicit integrator: drag_force = k_lin v, drag_torque = k_ang w
};

// A14-A15: lever-arm torques, sum (fp64, rounded to fp32 here and nowhere earlier), safety clamp
// (hydrodynamics_behavior.py:212-226).  A dry body gets exact zeros (A4: selects, not multiplies).
HYDRO_FN Wrench assemble_wrench(const Body& o, float mass)
{
    const double gx = o.drag_fx + o.lift_fx, gy = o.drag_fy + o.lift_fy, gz = o.drag_fz + o.lift_fz;   // act at the centre of pressure
    const double fx = gx + o.am_fx, fy = gy + o.am_fy, fz = o.buoy_z + (gz + o.am_fz);
    // tau = arm_b x (0,0,Fb) + arm_p x (F_drag + F_lift) + tau_drag + tau_am
    const double tx = o.armb_y * o.buoy_z + (o.armp_y * gz - o.armp_z * gy) + o.drag_tx + o.am_tx;
    const double ty = -o.armb_x * o.buoy_z + (o.armp_z * gx - o.armp_x * gz) + o.drag_ty + o.am_ty;
    const double tz = (o.armp_x * gy - o.armp_y * gx) + o.drag_tz + o.am_tz;
    // The clamp factor multiplies the finished sums - nothing cancels after it - so it is the one quantity evaluated
    // in fp32 (v_sqrt_f32 / v_rcp_f32, 1 ulp each): ~2e-7 on the results of the bodies it applies to (scale < 1).
    const float f_mag = fast_sqrt((float)(fx * fx + fy * fy + fz * fz));
    const float scale = fminf(1.0f, (mass * (float)kMaxAccel) * fast_rcp(f_mag + (float)kClampEps));
    Wrench w;
    w.fx = o.wet ? (float)fx * scale : 0.0f; w.fy = o.wet ? (float)fy * scale : 0.0f; w.fz = o.wet ? (float)fz * scale : 0.0f;
    w.tx = o.wet ? (float)tx * scale : 0.0f; w.ty = o.wet ? (float)ty * scale : 0.0f; w.tz = o.wet ? (float)tz * scale : 0.0f;
    w.k_lin = o.wet ? (float)o.lin_k * scale : 0.0f;
    w.k_ang = o.wet ? (float)o.ang_k * scale : 0.0f;
    return w;
}

// One body of the fused entry points: A13 (finite-difference acceleration from the previous-step velocity,
// hydrodynamics_behavior.py:196-202) + A1-A11 + A14-A15.  inv_dt = 1/dt in fp64 (dt is a double through the C ABI).
HYDRO_FN Wrench solve_wrench(const BodyIn& b, const float (&pv)[6], float mass, double rho, double g, double inv_dt, bool warp)
{
    const double ax = ((double)b.vx - (double)pv[0]) * inv_dt, ay = ((double)b.vy - (double)pv[1]) * inv_dt, az = ((double)b.vz - (double)pv[2]) * inv_dt;
    const double bx = ((double)b.wx - (double)pv[3]) * inv_dt, by = ((double)b.wy - (double)pv[4]) * inv_dt, bz = ((double)b.wz - (double)pv[5]) * inv_dt;
    return assemble_wrench(solve_body(b, ax, ay, az, bx, by, bz, rho, g, warp), mass);
}

// The same with the late inputs fetched by the caller's `load(anchor, pv, am_lin, am_ang, mass)` at the point of use
// (see solve_body_with): the previous-step velocity, the two added-mass coefficients and the mass are what the last
// third of the evaluation needs and nothing before it does.  Same expressions, same bits as solve_wrench.
template <class LateLoad>
HYDRO_FN Wrench solve_wrench_late(const BodyIn& b, LateLoad&& load, double rho, double g, double inv_dt, bool warp)
{
    float mass = 0.0f;
    const Body o = solve_body_with(b, [&](double anchor, double& ax, double& ay, double& az, double& bx, double& by, double& bz,
                                          float& am_lin, float& am_ang) {
        float pv[6];
        load(anchor, pv, am_lin, am_ang, mass);
        ax = ((double)b.vx - (double)pv[0]) * inv_dt; ay = ((double)b.vy - (double)pv[1]) * inv_dt; az = ((double)b.vz - (double)pv[2]) * inv_dt;
        bx = ((double)b.wx - (double)pv[3]) * inv_dt; by = ((double)b.wy - (double)pv[4]) * inv_dt; bz = ((double)b.wz - (double)pv[5]) * inv_dt;
    }, rho, g, warp);
    return assemble_wrench(o, mass);
}

// The calculator surface (calculate_hydrodynamic_forces, numba_hydrodynamics_wrapper.py:34-53): the eight vectors
// and the ratio as fp32, world-space centres.  A dry body returns zeros for everything, centres included (Numba,
// :277-279); the Warp twin reports cob = cop = the position, or the mean of whatever keypoints are wet (N6,
// warp_hydrodynamics.py:59-61,290).
struct Components {
    float ratio;
    float v[8][3];                  // buoyancy F, drag F, lift F, drag T, added-mass F, added-mass T, cob, cop
};
HYDRO_FN Components round_components(const Body& o, const BodyIn& b, bool warp)
{
    Components c;
    const bool w = o.wet;
    c.ratio = w ? (float)o.ratio : 0.0f;
    const double f[6][3] = {{0.0, 0.0, o.buoy_z}, {o.drag_fx, o.drag_fy, o.drag_fz}, {o.lift_fx, o.lift_fy, o.lift_fz},
                            {o.drag_tx, o.drag_ty, o.drag_tz}, {o.am_fx, o.am_fy, o.am_fz}, {o.am_tx, o.am_ty, o.am_tz}};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) c.v[k][a] = w ? (float)f[k][a] : 0.0f;
    const bool centres = w || warp;
    const double p[3] = {(double)b.px, (double)b.py, (double)b.pz};
    const double cb[3] = {o.armb_x, o.armb_y, o.armb_z};
    const double cp[3] = {w ? o.armp_x : o.armb_x, w ? o.armp_y : o.armb_y, w ? o.armp_z : o.armb_z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.v[6][a] = centres ? (float)(p[a] + cb[a]) : 0.0f;
        c.v[7][a] = centres ? (float)(p[a] + cp[a]) : 0.0f;
    }
    return c;
}

// Kinetic energy of one body (new functionality named by BASELINE.json north_star, absent from the reference:
// SURVEY.md 8e): translational 1/2 m |v|^2 and, with the box inertia I = m/12 diag(dy^2+dz^2, dx^2+dz^2, dx^2+dy^2) in
// the body frame, rotational 1/2 w_b.I.w_b with w_b = R^T w.  fp64 from the fp32 state; the rotation matrix is
// written exactly as in solve_body_with, so a kernel that evaluates both shares it.
HYDRO_FN void kinetic_energy(float fqx, float fqy, float fqz, float fqw, float fvx, float fvy, float fvz,
                             float fwx, float fwy, float fwz, float fdx, float fdy, float fdz, float fmass,
                             bool rotational, double& lin, double& rot)
{
    const double m = fmass;
    const double vx = fvx, vy = fvy, vz = fvz;
    lin = 0.5 * m * (vx * vx + vy * vy + vz * vz);
    rot = 0.0;
    if (rotational) {
        const double qx = fqx, qy = fqy, qz = fqz, qw = fqw;
        const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
        const double xx = qx * x2, yy = qy * y2, zz = qz * z2;
        const double sx = qw * x2, sy = qw * y2, sz = qw * z2;
        const double r00 = 1.0 - (yy + zz), r01 = __builtin_fma(qx, y2, -sz), r02 = __builtin_fma(qx, z2, sy);
        const double r10 = __builtin_fma(qx, y2, sz), r11 = 1.0 - (xx + zz), r12 = __builtin_fma(qy, z2, -sx);
        const double r20 = __builtin_fma(qx, z2, -sy), r21 = __builtin_fma(qy, z2, sx), r22 = 1.0 - (xx + yy);
        const double wx = fwx, wy = fwy, wz = fwz;
        const double bx = r00 * wx + r10 * wy + r20 * wz, by = r01 * wx + r11 * wy + r21 * wz, bz = r02 * wx + r12 * wy + r22 * wz;
        const double dx = fdx, dy = fdy, dz = fdz;
        const double k = m / 12.0;
        rot = 0.5 * k * ((dy * dy + dz * dz) * (bx * bx) + (dx * dx + dz * dz) * (by * by) + (dx * dx + dy * dy) * (bz * bz));
    }
}

}  // namespace hydro
