// Per-body hydrodynamic wrench arithmetic for the gfx950 kernels.
//
// One call = one rigid body, one physics step.  fp32, with an fp64 island (~120 of ~590 VALU
// instructions) for the quantities whose fp32 rounding would be amplified by a cancellation:
//   * the vertical extent / submersion numerator: `ratio = -z_min / (z_max - z_min)` cancels
//     catastrophically for barely-wet bodies (z_min = p_z - extent; SURVEY.md section 7 "hard parts");
//   * the body-frame flow direction R^T v (small components of edge-on faces);
//   * buoyancy + drag along z, summed before rounding, and the buoyancy torque: the largest term of the
//     wrench and the two that routinely cancel it (scene scalars rho, g arrive as doubles for the same reason);
//   * |q|^2 - 1, which the reference carries into terms that would otherwise cancel exactly.
// No MFMA: the path is elementwise per body.
//
// The model (what must come out) is the reference's
//   numba_hydrodynamics.py:9-314      (A1-A11: rotation, submersion + CoB,
//                                      CoP + projected area, hybrid drag, lift,
//                                      added mass)
//   hydrodynamics_behavior.py:196-226 (A13-A15: finite-difference acceleration,
//                                      lever-arm torques, sum, 500 m/s^2 clamp)
// The evaluation is restructured for the GPU (all closed forms checked against
// the fp64 oracle, tests/test_numerics_host.py and tests/test_parity_gpu.py):
//   * keypoint heights come from row 2 of R only:
//       z_ijk = p_z + i*e_x + j*e_y + k*e_z,  e_a = h_a * R[2][a],  i,j,k in {-1,0,1}
//     so z_min/max = p_z -/+ (|e_x|+|e_y|+|e_z|) and the 27 world points are
//     never formed (numba_hydrodynamics.py:271 builds them with a 3x27 GEMM);
//   * CoB and CoP are kept as BODY-RELATIVE lever arms (cob - p, cop - p):
//     the wrench only ever uses those differences (hydrodynamics_behavior.py:
//     212-214) and forming world-space points first loses ~|p|*2^-24 in fp32;
//   * at most one face per axis opposes the flow, selected by sign(R^T v_hat);
//   * sin(2*asin(d)) = 2 d sqrt((1-d)(1+d));  (axis x v_hat) = up|v_hat|^2 + d v_hat;
//   * the CoP-arm drag torque is formed in the body frame, where its zero (all three faces wet) is structural.
// N1 completion (speed <= 1e-6 -> area 0, CoP = CoB) as in oracle/hydro_oracle.py.
//
// This header is compiled for the device by hipcc and, for the CPU-side
// numerics study only (tests/host_emul), by the host compiler; the product
// never calls the host instantiation.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HYDRO_FN __host__ __device__ __forceinline__
#else
#define HYDRO_FN static inline __attribute__((always_inline))
#endif

namespace hydro {

constexpr float kSpeedEps = 1e-6f;      // numba_hydrodynamics.py:118,156,170,192,286
constexpr float kLowSpeed = 0.2f;       // :154
constexpr float kInvLowSpeed = 5.0f;
constexpr float kDryEps = 1e-9f;        // :192,225,277
constexpr float kAreaEps = 1e-6f;       // :140
constexpr float kHeightEps = 1e-6f;     // :92
constexpr float kAxisEps = 1e-6f;       // :210
constexpr float kMaxAccel = 500.0f;     // hydrodynamics_behavior.py:221
constexpr float kClampEps = 1e-6f;      // hydrodynamics_behavior.py:224

// Single-instruction transcendental forms on the device (v_sqrt_f32 / v_rcp_f32,
// 1 ulp); plain libm on the host instantiation.
HYDRO_FN float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
HYDRO_FN float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// Issue priority of this wavefront over the others on its SIMD (device only).
HYDRO_FN void raise_priority()
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HYDRO_NO_SETPRIO)
    __builtin_amdgcn_s_setprio(3);
#endif
}

// True when `x` holds in ANY lane of the wavefront (device); the host instantiation has one "lane".
HYDRO_FN bool any_lane(bool x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(x) != 0;
#else
    return x;
#endif
}

HYDRO_FN uint32_t float_bits(float x) { uint32_t u; __builtin_memcpy(&u, &x, sizeof u); return u; }
// mask = (mask << 1) | signbit(z): one v_alignbit_b32 on the device
HYDRO_FN uint32_t shift_in_sign(uint32_t mask, float z) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(mask, float_bits(z), 31);
#else
    return (mask << 1) | (float_bits(z) >> 31);
#endif
}
// bits of the 27-bit keypoint mask (point p = 9 i + 3 j + k sits at bit 26 - p) whose lattice index
// along `axis` (0: i, 1: j, 2: k) equals `value` (0, 1, 2 for -1, 0, +1)
constexpr uint32_t lattice_mask(int axis, int value) {
    uint32_t m = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const int idx = axis == 0 ? i : (axis == 1 ? j : k);
                if (idx == value) m |= 1u << (26 - (9 * i + 3 * j + k));
            }
    return m;
}

struct BodyIn {
    float px, py, pz;
    float qx, qy, qz, qw;           // unit quaternion xyzw, NOT normalised (N7)
    float vx, vy, vz;
    float wx, wy, wz;
    float ax, ay, az;               // linear acceleration
    float bx, by, bz;               // angular acceleration
    float pvx, pvy, pvz;            // previous-step linear / angular velocity: read by wrench_fp64 only
    float pwx, pwy, pwz;            //   (the fused entry points; component mode gets accelerations and leaves them unset)
    float dimx, dimy, dimz;
    float cd_lin, cd_ang, damp_lin, damp_ang, lift, am_lin, am_ang;
};

struct BodyOut {
    float ratio;                    // 0 when dry
    float buoy_z;                   // buoyancy force is (0,0,buoy_z)
    float drag_fx, drag_fy, drag_fz;
    float lift_fx, lift_fy, lift_fz;
    float drag_tx, drag_ty, drag_tz;
    float am_fx, am_fy, am_fz;
    float am_tx, am_ty, am_tz;
    float armb_x, armb_y, armb_z;   // cob - p   (0 when dry)
    float armp_x, armp_y, armp_z;   // cop - p   (0 when dry)
    float dragarm_tx, dragarm_ty, dragarm_tz;   // (cop - p) x drag_force, cancellation-free form
    float tbx, tby;                 // buoyancy torque (cob - p) x (0,0,B), evaluated in fp64 (0 when dry)
    float fz_core;                  // buoyancy + drag force along z, summed in fp64 (0 when dry)
    float lin_k, ang_k;             // drag_force = lin_k * v, drag_torque = ang_k * w   (both <= 0; 0 when dry)
    uint32_t wetmask;               // 27 keypoint sign bits (bit 26 - (9 i + 3 j + k) set: lattice point (i,j,k) is below the surface)
    bool wet;
};

// A1-A11.  rho, g are scene scalars (hydrodynamics_config.json:2-5 "globals").
// They arrive as doubles (the reference passes Python floats, numba_hydrodynamics_wrapper.py:9-10): 9.81 is
// not an fp32 number, and rounding it costs 4e-8 of the buoyancy before any arithmetic has happened.
// `warp` (uniform over a launch) selects the semantics of the reference's Warp twin where it differs from
// the Numba path (include/hydro.h HYDRO_SEM_WARP); the default is Numba.
// ZERO_DRY = false leaves the outputs of a dry body unselected (whatever the arithmetic produced): for
// callers that go straight to assemble_wrench, which zeroes the six results of a dry body itself - 6 selects
// instead of ~30.
template <bool ZERO_DRY = true>
HYDRO_FN BodyOut solve_body(const BodyIn& b, double rho64, double g64, bool warp = false)
{
    const float rho = (float)rho64, g = (float)g64;
    BodyOut o;

    // ---- A1: rotation matrix, fp32 (numba_hydrodynamics.py:14-49) ----
    const float x2 = b.qx + b.qx, y2 = b.qy + b.qy, z2 = b.qz + b.qz;
    const float xx = b.qx * x2, xy = b.qx * y2, xz = b.qx * z2;
    const float yy = b.qy * y2, yz = b.qy * z2, zz = b.qz * z2;
    const float wx = b.qw * x2, wy = b.qw * y2, wz = b.qw * z2;
    const float r00 = 1.0f - (yy + zz), r01 = xy - wz, r02 = xz + wy;
    const float r10 = xy + wz, r11 = 1.0f - (xx + zz), r12 = yz - wx;

    // ---- row 2 and the vertical extent in fp64 (conditioning, see header) ----
    const double dqx = b.qx, dqy = b.qy, dqz = b.qz, dqw = b.qw;
    // half of row 2 (t2a = R[2][a] / 2): the half-extent h_a = dim_a / 2 then needs no conversion of its own
    // (e_a = h_a R[2][a] = dim_a t2a) and the factor 2 is applied to the fp32 copies, where it is exact
    const double t20 = dqx * dqz - dqw * dqy;
    const double t21 = dqy * dqz + dqw * dqx;
    const double t22 = 0.5 - (dqx * dqx + dqy * dqy);
    const double ddx = b.dimx, ddy = b.dimy, ddz = b.dimz;
    const float hx = 0.5f * b.dimx, hy = 0.5f * b.dimy, hz = 0.5f * b.dimz;
    const double dex = ddx * t20, dey = ddy * t21, dez = ddz * t22;
    const double extent = fabs(dex) + fabs(dey) + fabs(dez);
    const double zlo = (double)b.pz - extent;           // lowest keypoint  (z_min)
    const double zhi = (double)b.pz + extent;           // highest keypoint (z_max)
    const float r20 = 2.0f * (float)t20, r21 = 2.0f * (float)t21, r22 = 2.0f * (float)t22;
    const float ex = (float)dex, ey = (float)dey, ez = (float)dez;
    // The quaternion is used as given, never normalised (N7).  With e = |q|^2 - 1 the matrix above is
    // R = (1+e) R^ - e I for the true rotation R^, which gives the EXACT identities
    //     R R^T = R^T R = (1 + 2e) I - e (R + R^T),        |R[:,2]|^2 - 1 = 2 e (1 - R22).
    // e ~ 1e-7 for an fp32-rounded unit quaternion: invisible to fp32 arithmetic, but the fp64
    // reference carries it into terms that otherwise cancel exactly (CoP lever arm x drag, 1 - d^2 in
    // the lift coefficient, R R^T a in the added mass), so e is evaluated here in fp64 and those
    // terms are written with the identities - exact for ANY quaternion, unit or not.
    const double qerr = ((dqx * dqx + dqy * dqy) + (dqz * dqz + dqw * dqw)) - 1.0;
    const float qe = (float)qerr;

    // ---- A3: submersion ratio (numba_hydrodynamics.py:86-96) ----
    const bool dry_by_extent = zlo >= 0.0;
    const bool fully_in = zhi <= 0.0;
    const float height = (float)(extent + extent);
    float ratio = fminf(1.0f, (float)(-zlo) * fast_rcp(height));
    if (height < kHeightEps) ratio = 1.0f;              // z_lo < 0 is known here
    if (fully_in) ratio = 1.0f;
    if (dry_by_extent) ratio = 0.0f;
    const bool wet = ratio > kDryEps;
    o.wet = wet;
    // ---- A5 in fp64: buoyancy (:282) from an fp64 ratio (one Newton step on -z_lo / height) ----
    // Buoyancy is routinely the largest term of the wrench, and two other terms routinely cancel it: drag
    // along z (a body sinking or rising near its terminal velocity; 180x cancellation observed) and the
    // drag torques against its lever-arm torque.  The fp32 roundings of the partners (1e-7 each, 4e-8 from
    // rounding g = 9.81 alone) are amplified by the cancellation ratio, so buoyancy, the z-drag and the
    // buoyancy torque are evaluated in fp64 from the raw inputs and rounded AFTER they have been summed
    // (below: fz_core, tbx, tby, lift_base).  ~80 more fp64 instructions per body; what they cost on MI355X is
    // measured in DESIGN.md section 5 (nothing at 4M bodies or with fp32 coefficients, ~5 % in the 1M fp16 case).
    const double vol64 = (ddx * ddy) * ddz;
    double ratio64 = (double)ratio + ((-zlo) - (double)ratio * (extent + extent)) * (double)fast_rcp(height);
    if (ratio >= 1.0f) ratio64 = 1.0;                   // clamped, degenerate height or fully in
    const double buoy64 = (rho64 * g64) * (ratio64 * vol64);

    // ---- A3: centre of buoyancy from integer lattice sums (:69-84,99-103) ----
    // z_ijk < 0 for lattice index (i,j,k); S_a = sum of index a over the wet points.  The test is the
    // sign bit of z (p_z + 0.0f turns an input of -0.0 into +0.0, after which no sum below can be
    // -0.0).  The 27 sign bits are funnel-shifted into ONE mask register (one v_alignbit_b32 per
    // keypoint); the count and the three index sums are then popcounts of that mask against
    // compile-time masks - no per-keypoint compare / select / float accumulation.
    const float pzc = b.pz + 0.0f;
    const float zi[3] = {pzc - ex, pzc, pzc + ex};
    uint32_t wetmask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float zij[3] = {zi[i] - ey, zi[i], zi[i] + ey};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            wetmask = shift_in_sign(wetmask, zij[j] - ez);
            wetmask = shift_in_sign(wetmask, zij[j]);
            wetmask = shift_in_sign(wetmask, zij[j] + ez);
        }
    }
    o.wetmask = wetmask;
    const float cnt = (float)__builtin_popcount(wetmask);
    const float s_i = (float)(__builtin_popcount(wetmask & lattice_mask(0, 2)) - __builtin_popcount(wetmask & lattice_mask(0, 0)));
    const float s_j = (float)(__builtin_popcount(wetmask & lattice_mask(1, 2)) - __builtin_popcount(wetmask & lattice_mask(1, 0)));
    const float s_k = (float)(__builtin_popcount(wetmask & lattice_mask(2, 2)) - __builtin_popcount(wetmask & lattice_mask(2, 0)));
    const bool partial = !dry_by_extent && !fully_in && cnt > 0.0f;
    const float inv_cnt = partial ? fast_rcp(cnt) : 0.0f;
    const float lbx = hx * s_i * inv_cnt, lby = hy * s_j * inv_cnt, lbz = hz * s_k * inv_cnt;   // body frame
    float armb_x = r00 * lbx + r01 * lby + r02 * lbz;
    float armb_y = r10 * lbx + r11 * lby + r12 * lbz;
    float armb_z = r20 * lbx + r21 * lby + r22 * lbz;
    // buoyancy torque (cob - p) x (0,0,B) = (arm_y B, -arm_x B, 0): the HORIZONTAL lever arm times the largest
    // force.  For a near-upright body arm_x, arm_y are small components of the rotated lattice mean
    // (|arm| ~ h_z): the fp32 product R l only delivers them to ~1e-7 |arm| absolute (4e-6 relative
    // observed), and so does an fp64 rotation of the fp32-rounded l (8e-6 of the torque at 0.3 degrees of tilt,
    // 1e-5 of the metric's floor for a body floating exactly upright).  So the lattice mean itself is formed in
    // fp64 - l = dim_a (S_a / 2 cnt), 1/cnt by one Newton step on the fp32 reciprocal - and rotated in fp64:
    //     R l = l + 2 (w t + q x t),  t = q x l     (x, y components only).
    float tbx, tby;
    {
        const double r0 = (double)inv_cnt, half_inv_cnt = r0 * (1.0 - (0.5 * (double)cnt) * r0);   // 1 / (2 cnt)
        const double l0 = ddx * ((double)s_i * half_inv_cnt), l1 = ddy * ((double)s_j * half_inv_cnt), l2 = ddz * ((double)s_k * half_inv_cnt);
        const double bt0 = dqy * l2 - dqz * l1, bt1 = dqz * l0 - dqx * l2, bt2 = dqx * l1 - dqy * l0;
        const double arm64x = l0 + 2.0 * (dqw * bt0 + (dqy * bt2 - dqz * bt1));
        const double arm64y = l1 + 2.0 * (dqw * bt1 + (dqz * bt0 - dqx * bt2));
        tbx = (float)(arm64y * buoy64);
        tby = (float)(-arm64x * buoy64);
    }

    // ---- A5: buoyancy (:282) ----
    const float volume = b.dimx * b.dimy * b.dimz;
    const float buoy_z = rho * (ratio * volume) * g;

    // ---- A6: speed and direction (:285-289) ----
    // |v|, 1/|v| and v_hat correctly rounded from fp64 (Newton steps on the fp32 v_sqrt / v_rcp seeds): the speed
    // enters the drag twice (1/2 rho s Cd A * s) and 1/|v| scales both v_hat and the body-frame direction u, so the
    // 1-ulp errors of the device's sqrt and rcp (which the host instantiation does not have) showed up 3x in the
    // tail of the GPU soak.  7 more fp64-class instructions.
    const double dvx = b.vx, dvy = b.vy, dvz = b.vz;
    const double v2_64 = dvx * dvx + dvy * dvy + dvz * dvz;
    const float speed0 = fast_sqrt(b.vx * b.vx + b.vy * b.vy + b.vz * b.vz);
    const bool moving = speed0 > kSpeedEps;
    const double rinv = moving ? (double)fast_rcp(speed0) : 0.0;
    const double s0 = v2_64 * rinv;
    const double speed64 = s0 + (v2_64 - s0 * s0) * (0.5 * rinv);
    const double inv64 = rinv * (2.0 - speed64 * rinv);
    const float speed = (float)speed64;
    const float inv_speed = (float)inv64;
    const float dx = (float)(dvx * inv64), dy = (float)(dvy * inv64), dz = (float)(dvz * inv64);   // v_hat (0 at rest)

    // ---- A7: projected area + centre of pressure (:108-143) ----
    // u = R^T v_hat; face (axis a, sign s) has alignment -s*u_a and centre height p_z + s*e_a.
    // In fp64 from the raw inputs: a face nearly edge-on to the flow has |u_a| << 1, and the fp32 dot
    // product R^T v_hat only delivers u_a to ~1e-7 ABSOLUTE (rounded R entries, rounded v_hat, rounded
    // sum).  When that face is the only wet one opposing the flow, the projected area, the CoP lever arm
    // and the lift all inherit the relative error (7e-6 observed at |u_a| ~ 0.01).  18 fp64 operations
    // (on their own measured free on MI355X: 23.15 -> 23.18 us at 1M bodies).
    // u = R^T v = v - 2 (w t - q x t),  t = q x v   (same polynomial in q as the matrix form, any |q|)
    const double tx_ = dqy * dvz - dqz * dvy, ty_ = dqz * dvx - dqx * dvz, tz_ = dqx * dvy - dqy * dvx;
    const double gx_ = (dqy * tz_ - dqz * ty_) - dqw * tx_;
    const double gy_ = (dqz * tx_ - dqx * tz_) - dqw * ty_;
    const double gz_ = (dqx * ty_ - dqy * tx_) - dqw * tz_;
    const double urx = dvx + 2.0 * gx_, ury = dvy + 2.0 * gy_, urz = dvz + 2.0 * gz_;      // R^T v
    const float ux = (float)(urx * inv64), uy = (float)(ury * inv64), uz = (float)(urz * inv64);
    const float sx = (ux < 0.0f) ? 1.0f : -1.0f;        // sign of the face opposing the flow
    const float sy = (uy < 0.0f) ? 1.0f : -1.0f;
    const float sz = (uz < 0.0f) ? 1.0f : -1.0f;
    const float fax = ((ux != 0.0f) && (b.pz + sx * ex < 0.0f)) ? fabsf(ux) * (b.dimy * b.dimz) : 0.0f;
    const float fay = ((uy != 0.0f) && (b.pz + sy * ey < 0.0f)) ? fabsf(uy) * (b.dimx * b.dimz) : 0.0f;
    const float faz = ((uz != 0.0f) && (b.pz + sz * ez < 0.0f)) ? fabsf(uz) * (b.dimx * b.dimy) : 0.0f;
    const bool cx_ = fax != 0.0f, cy_ = fay != 0.0f, cz_ = faz != 0.0f;
    const float area = fax + fay + faz;                 // 0 at rest (u = 0): N1 completion
    // buoyancy + drag along z in fp64, rounded after the sum.  s A needs no division:
    // s A = sum_a |(R^T v)_a| area_a with the un-normalised fp64 R^T v; |v| = speed64 from above.
    float fz_core, lift_base;
    {
        const double sA64 = (cx_ ? fabs(urx) * (ddy * ddz) : 0.0) + (cy_ ? fabs(ury) * (ddx * ddz) : 0.0)
                          + (cz_ ? fabs(urz) * (ddx * ddy) : 0.0);
        const double quad64 = moving ? (0.5 * rho64) * ((double)b.cd_lin * sA64) : 0.0;
        const double scale64 = (speed < kLowSpeed) ? speed64 * 5.0 : 1.0;
        const double link64 = (quad64 + (double)b.damp_lin * scale64) * ratio64;
        fz_core = (float)(buoy64 - link64 * dvz);
        // 1/2 rho s^2 A ratio for the lift (:201), from the same fp64 pieces: the lift-arm torque routinely cancels
        // the buoyancy / drag-arm torques 50-100x, and the fp32 chain of seven products was its weakest link (4e-7)
        lift_base = (float)(((0.5 * rho64) * (speed64 * sA64)) * ratio64);
    }
    const bool has_area = area > kAreaEps;
    const float inv_area = has_area ? fast_rcp(area) : 0.0f;
    // body-frame CoP arm: s_a h_a (|u_a| area_a) / A = -(V/2A) u_a on the axes that count (h_a area_a = V/2)
    const float aax = cx_ ? ux : 0.0f, aay = cy_ ? uy : 0.0f, aaz = cz_ ? uz : 0.0f;     // a = W u
    const float nhva = -0.5f * (b.dimx * b.dimy * b.dimz) * inv_area;                    // -(V/2A)
    const float lpx = nhva * aax, lpy = nhva * aay, lpz = nhva * aaz;
    float armp_x = r00 * lpx + r01 * lpy + r02 * lpz;
    float armp_y = r10 * lpx + r11 * lpy + r12 * lpz;
    float armp_z = r20 * lpx + r21 * lpy + r22 * lpz;
    if (!has_area) { armp_x = armb_x; armp_y = armb_y; armp_z = armb_z; }     // cop = cob (:115,140)
    // arm_p x v_hat without cancellation.  h_a * area_a = V/2 on every axis, so
    //     arm_p = -(V/2A) R a,   a = W u,   u = R^T v_hat,   W = diag(face of axis a opposes the flow and is wet)
    // (above), which is parallel to v_hat (no torque from drag) when all three opposing faces are wet.  Take the cross
    // product in the BODY frame, where that zero is structural:
    //     a x u = ((w_y - w_z) u_y u_z, (w_z - w_x) u_z u_x, (w_x - w_y) u_x u_y)
    // and carry the non-orthogonality of the reference's matrix exactly (R = (1+e) R^ - e I, N7):
    //     (R a) x v_hat = R y + e (y - a x v_hat),      y = (a x u + e (a x v_hat)) / (1 + e)
    // (an identity in e, checked to 1e-15 for |q| in [0.85, 1.1]).  Every product is between quantities of
    // full relative accuracy (u comes from the fp64 island) - nothing cancels, for any W and any |q|.
    const bool cx = fax != 0.0f, cy = fay != 0.0f, cz = faz != 0.0f;
    const float pyz = uy * uz, pzx = uz * ux, pxy = ux * uy;
    const float axu_x = (cy == cz) ? 0.0f : (cy ? pyz : -pyz);
    const float axu_y = (cz == cx) ? 0.0f : (cz ? pzx : -pzx);
    const float axu_z = (cx == cy) ? 0.0f : (cx ? pxy : -pxy);
    const float avx = aay * dz - aaz * dy, avy = aaz * dx - aax * dz, avz = aax * dy - aay * dx;   // a x v_hat
    const float inv1pe = fast_rcp(1.0f + qe);
    const float y0 = (axu_x + qe * avx) * inv1pe, y1 = (axu_y + qe * avy) * inv1pe, y2_ = (axu_z + qe * avz) * inv1pe;
    const float rax = (r00 * y0 + r01 * y1 + r02 * y2_) + qe * (y0 - avx);                // (R a) x v_hat
    const float ray = (r10 * y0 + r11 * y1 + r12 * y2_) + qe * (y1 - avy);
    const float raz = (r20 * y0 + r21 * y1 + r22 * y2_) + qe * (y2_ - avz);
    float pxv_x = nhva * rax, pxv_y = nhva * ray, pxv_z = nhva * raz;              // arm_p x v_hat
    if (!has_area) {
        pxv_x = armb_y * dz - armb_z * dy; pxv_y = armb_z * dx - armb_x * dz; pxv_z = armb_x * dy - armb_y * dx;
    }

    // ---- A8: hybrid drag (:146-182).  quad = -(1/2 rho s^2 Cd A) v_hat = -(1/2 rho s Cd A) v ----
    const float half_rho = 0.5f * rho;
    const float lin_quad = moving ? half_rho * speed * b.cd_lin * area : 0.0f;
    const float lin_scale = (speed < kLowSpeed) ? speed * kInvLowSpeed : 1.0f;
    const float lin_k = -(lin_quad + b.damp_lin * lin_scale) * ratio;
    const float wspeed = fast_sqrt(b.wx * b.wx + b.wy * b.wy + b.wz * b.wz);
    const float ang_quad = (wspeed > kSpeedEps) ? half_rho * wspeed * b.cd_ang * volume : 0.0f;
    const float ang_scale = (wspeed < kLowSpeed) ? wspeed * kInvLowSpeed : 1.0f;
    const float ang_k = -(ang_quad + b.damp_ang * ang_scale) * ratio;

    // ---- A9: lift (:185-217) ----
    // up = R[:,2]; d = clamp(-up.v_hat); C_L = sin(2 asin d) = 2 d sqrt((1-d)(1+d));
    // dir = (v_hat x up) x v_hat / |v_hat x up| = (up |v_hat|^2 + d_raw v_hat) / |v_hat x up|
    const float d_raw = -uz;                                        // up . v_hat = (R^T v_hat)_z
    // 1 - d^2 is taken from |v_hat x up|^2 = |up|^2 - d^2 (no cancellation as |d| -> 1):
    //     sqrt(1 - d^2) / |axis| = sqrt(max(0, 1 - eta / |axis|^2)),  eta = |up|^2 - 1 = 2 e (1 - R22)  (exact).
    const float dcl = fminf(1.0f, fmaxf(-1.0f, d_raw));
    const float axx = dy * r22 - dz * r12, axy = dz * r02 - dx * r22, axz = dx * r12 - dy * r02;
    const float n_axis2 = axx * axx + axy * axy + axz * axz;
    const bool lift_on = !(speed < kSpeedEps) && !(n_axis2 < kAxisEps * kAxisEps);   // |axis| < 1e-6 (:210), no sqrt needed
    const float eta = 2.0f * qe * (1.0f - r22);
    const float clamp_on = (fabsf(d_raw) < 1.0f) ? 1.0f : 0.0f;    // |d| >= 1 -> asin(+-1): C_L = sin(+-pi) = 0
    const float c_l_over_n = 2.0f * dcl * clamp_on * fast_sqrt(fmaxf(0.0f, 1.0f - eta * fast_rcp(n_axis2)));
    const float vhat2 = dx * dx + dy * dy + dz * dz;
    const float lift_k = lift_on ? lift_base * (c_l_over_n * b.lift) : 0.0f;

    // ---- A10: added mass (:220-253; diagonal of numba_hydrodynamics_wrapper.py:101-112) ----
    const float rv = volume * rho;
    const float m_lin = rv * b.am_lin;
    const float m_ang = rv * b.am_ang;
    const float d2x = b.dimx * b.dimx, d2y = b.dimy * b.dimy, d2z = b.dimz * b.dimz;
    // linear part: the added mass is isotropic (m_lin on all three axes), so the two rotations the
    // reference performs collapse:  R (m_lin R^T a) = m_lin (R R^T) a = m_lin ((1+2e) a - e (R + R^T) a).
    const float s01 = r01 + r10, s02 = r02 + r20, s12 = r12 + r21;                 // R + R^T (symmetric)
    const float sax = 2.0f * r00 * b.ax + s01 * b.ay + s02 * b.az;
    const float say = s01 * b.ax + 2.0f * r11 * b.ay + s12 * b.az;
    const float saz = s02 * b.ax + s12 * b.ay + 2.0f * r22 * b.az;
    const float one2e = 1.0f + 2.0f * qe;
    const float bbx = r00 * b.bx + r10 * b.by + r20 * b.bz;         // R^T alpha
    const float bby = r01 * b.bx + r11 * b.by + r21 * b.bz;
    const float bbz = r02 * b.bx + r12 * b.by + r22 * b.bz;
    const float kf = -m_lin * ratio;
    const float kt = -m_ang * ratio;
    const float tlx = kt * (d2y + d2z) * bbx, tly = kt * (d2x + d2z) * bby, tlz = kt * (d2x + d2y) * bbz;
    float am_fx = kf * (one2e * b.ax - qe * sax), am_fy = kf * (one2e * b.ay - qe * say), am_fz = kf * (one2e * b.az - qe * saz);
    float am_tx = r00 * tlx + r01 * tly + r02 * tlz;
    float am_ty = r10 * tlx + r11 * tly + r12 * tlz;
    float am_tz = r20 * tlx + r21 * tly + r22 * tlz;
    if (warp) {
        // N3: the Warp twin takes the world accelerations into the "local" frame with quat_rotate(q, .) = R
        // (warp_hydrodynamics.py:216-217) where Numba uses R^T (numba_hydrodynamics.py:229-230), and comes
        // back with R in both (:229-230 / :247-248):  F = R (-M (R a)).  Reproduced as written.
        const float alx = r00 * b.ax + r01 * b.ay + r02 * b.az;
        const float aly = r10 * b.ax + r11 * b.ay + r12 * b.az;
        const float alz = r20 * b.ax + r21 * b.ay + r22 * b.az;
        am_fx = kf * (r00 * alx + r01 * aly + r02 * alz);
        am_fy = kf * (r10 * alx + r11 * aly + r12 * alz);
        am_fz = kf * (r20 * alx + r21 * aly + r22 * alz);
        const float wlx = kt * (d2y + d2z) * (r00 * b.bx + r01 * b.by + r02 * b.bz);
        const float wly = kt * (d2x + d2z) * (r10 * b.bx + r11 * b.by + r12 * b.bz);
        const float wlz = kt * (d2x + d2y) * (r20 * b.bx + r21 * b.by + r22 * b.bz);
        am_tx = r00 * wlx + r01 * wly + r02 * wlz;
        am_ty = r10 * wlx + r11 * wly + r12 * wlz;
        am_tz = r20 * wlx + r21 * wly + r22 * wlz;
    }

    // ---- A4: dry bodies return zeros for every output (:277-279) ----
    // (selects, not multiplies: a dry body must give exact zeros whatever the rest evaluated to)
#define HYDRO_LIVE(x) ((ZERO_DRY && !wet) ? 0.0f : (x))
    o.ratio = HYDRO_LIVE(ratio);
    o.buoy_z = HYDRO_LIVE(buoy_z);
    o.drag_fx = HYDRO_LIVE(lin_k * b.vx); o.drag_fy = HYDRO_LIVE(lin_k * b.vy); o.drag_fz = HYDRO_LIVE(lin_k * b.vz);
    o.drag_tx = HYDRO_LIVE(ang_k * b.wx); o.drag_ty = HYDRO_LIVE(ang_k * b.wy); o.drag_tz = HYDRO_LIVE(ang_k * b.wz);
    o.lift_fx = HYDRO_LIVE(lift_k * (r02 * vhat2 + d_raw * dx));
    o.lift_fy = HYDRO_LIVE(lift_k * (r12 * vhat2 + d_raw * dy));
    o.lift_fz = HYDRO_LIVE(lift_k * (r22 * vhat2 + d_raw * dz));
    o.am_fx = HYDRO_LIVE(am_fx); o.am_fy = HYDRO_LIVE(am_fy); o.am_fz = HYDRO_LIVE(am_fz);
    o.am_tx = HYDRO_LIVE(am_tx); o.am_ty = HYDRO_LIVE(am_ty); o.am_tz = HYDRO_LIVE(am_tz);
    o.fz_core = HYDRO_LIVE(fz_core); o.tbx = HYDRO_LIVE(tbx); o.tby = HYDRO_LIVE(tby);
    o.lin_k = HYDRO_LIVE(lin_k); o.ang_k = HYDRO_LIVE(ang_k);
    // N6: a dry body's centres are zeros in Numba (:277-279); the Warp twin reports cob (the position, or the
    // mean of whatever keypoints are wet) and cop = cob (warp_hydrodynamics.py:59-61,290) - component mode only,
    // every force is zero either way.
    const bool arms = wet || warp || !ZERO_DRY;
    o.armb_x = arms ? armb_x : 0.0f; o.armb_y = arms ? armb_y : 0.0f; o.armb_z = arms ? armb_z : 0.0f;
    o.armp_x = (wet || !ZERO_DRY) ? armp_x : o.armb_x; o.armp_y = (wet || !ZERO_DRY) ? armp_y : o.armb_y;
    o.armp_z = (wet || !ZERO_DRY) ? armp_z : o.armb_z;
    const float ks = lin_k * speed;                                 // drag_force = ks * v_hat
    o.dragarm_tx = HYDRO_LIVE(ks * pxv_x); o.dragarm_ty = HYDRO_LIVE(ks * pxv_y); o.dragarm_tz = HYDRO_LIVE(ks * pxv_z);
#undef HYDRO_LIVE
    return o;
}

struct Wrench {
    float fx, fy, fz, tx, ty, tz;
    float scale;                    // the clamp factor that was applied
    bool ill;                       // assemble_wrench: passed the 1-norm screen; after solve_wrench: was re-evaluated in fp64
};

// A body whose net force or net torque is more than kCancelGate times smaller than the terms it is the sum of
// (2-norms) is re-evaluated in fp64 by wrench_fp64.  Every term of the fp32 evaluation is good to 1-2.5e-7 of
// ITSELF, so up to the gate the sum is good to 12 * 2.5e-7 = 3e-6 of itself, and past it fp32 terms cannot deliver
// 1e-5: 70-350x cancellations were the only margin-gated bodies above 1e-5 in 1.38e9 evaluations of round 1.
// The test runs in two stages: assemble_wrench screens with 1-norms (a handful of |x| additions on the path every
// body takes; ||x||_2 <= ||x||_1 <= sqrt(3) ||x||_2, so a body past the gate always passes the screen at
// gate / sqrt(3)), and only wavefronts in which some lane passed the screen evaluate the 2-norm form
// (cancels_2norm, ~3 % of the wavefronts of the bench scenes); ~0.5 % of them go on to the fp64 pass.
#ifndef HYDRO_CANCEL_GATE
#define HYDRO_CANCEL_GATE 12.0f       // build-time knob for A/B measurements (scripts/ab_variants.py)
#endif
constexpr float kCancelGate = HYDRO_CANCEL_GATE;
constexpr float kCancelScreen = kCancelGate * 0.57735026f;      // gate / sqrt(3)
constexpr float kCancelFloor = 1e-3f;   // forces below 1e-3 rho g V (torques: x the longest edge) count as zero

// A14-A15: lever-arm torques, sum, safety clamp (hydrodynamics_behavior.py:212-226).
HYDRO_FN Wrench assemble_wrench(const BodyOut& o, float mass, float weight_scale = 0.0f, float lmax = 0.0f)
{
    const float fx = (o.drag_fx + o.lift_fx) + o.am_fx;
    const float fy = (o.drag_fy + o.lift_fy) + o.am_fy;
    const float fz = o.fz_core + (o.lift_fz + o.am_fz);            // fz_core = buoyancy + drag_z, summed in fp64
    // tau = arm_b x (0,0,Fb) + arm_p x F_drag + arm_p x F_lift + tau_drag + tau_am
    const float lax = o.armp_y * o.lift_fz - o.armp_z * o.lift_fy;
    const float lay = o.armp_z * o.lift_fx - o.armp_x * o.lift_fz;
    const float laz = o.armp_x * o.lift_fy - o.armp_y * o.lift_fx;
    const float tx = o.tbx + (o.dragarm_tx + lax + o.drag_tx + o.am_tx);
    const float ty = o.tby + (o.dragarm_ty + lay + o.drag_ty + o.am_ty);
    const float tz = o.dragarm_tz + laz + o.drag_tz + o.am_tz;
    const float f_mag = fast_sqrt(fx * fx + fy * fy + fz * fz);
    const float scale = fminf(1.0f, mass * kMaxAccel * fast_rcp(f_mag + kClampEps));
    // conditioning of the two sums: 1-norm of the terms against the 1-norm of the result (fz_core counts as ONE
    // term: buoyancy and z-drag were summed in fp64).  weight_scale = 1e-3 rho g V is the size below which the
    // parity metric and the physics treat a force as zero.
    const float sum_f = (fabsf(o.drag_fx) + fabsf(o.drag_fy) + fabsf(o.fz_core))
                      + (fabsf(o.lift_fx) + fabsf(o.lift_fy) + fabsf(o.lift_fz))
                      + (fabsf(o.am_fx) + fabsf(o.am_fy) + fabsf(o.am_fz));
    const float sum_t = (fabsf(o.tbx) + fabsf(o.tby)) + (fabsf(o.dragarm_tx) + fabsf(o.dragarm_ty) + fabsf(o.dragarm_tz))
                      + (fabsf(lax) + fabsf(lay) + fabsf(laz)) + (fabsf(o.drag_tx) + fabsf(o.drag_ty) + fabsf(o.drag_tz))
                      + (fabsf(o.am_tx) + fabsf(o.am_ty) + fabsf(o.am_tz));
    const float net_f = fmaxf(fabsf(fx) + fabsf(fy) + fabsf(fz), weight_scale);
    const float net_t = fmaxf(fabsf(tx) + fabsf(ty) + fabsf(tz), weight_scale * lmax);
    // A4: a dry body gets exact zeros (selects, not multiplies - whatever the rest evaluated to)
    Wrench w;
    w.fx = o.wet ? fx * scale : 0.0f; w.fy = o.wet ? fy * scale : 0.0f; w.fz = o.wet ? fz * scale : 0.0f;
    w.tx = o.wet ? tx * scale : 0.0f; w.ty = o.wet ? ty * scale : 0.0f; w.tz = o.wet ? tz * scale : 0.0f;
    w.scale = scale;
    w.ill = o.wet && (sum_f > kCancelScreen * net_f || sum_t > kCancelScreen * net_t);
    return w;
}

// Second stage of the cancellation test (see kCancelGate): sum of the 2-norms of the terms against the 2-norm of
// their sum, for the force (buoyancy + drag - summed in fp64 where they cancel along z - lift, added mass) and for the
// torque (buoyancy arm, drag arm, lift arm, angular drag, added mass).
HYDRO_FN bool cancels_2norm(const BodyOut& o, float weight_scale, float lmax)
{
    const float fx = (o.drag_fx + o.lift_fx) + o.am_fx, fy = (o.drag_fy + o.lift_fy) + o.am_fy, fz = o.fz_core + (o.lift_fz + o.am_fz);
    const float lax = o.armp_y * o.lift_fz - o.armp_z * o.lift_fy;
    const float lay = o.armp_z * o.lift_fx - o.armp_x * o.lift_fz;
    const float laz = o.armp_x * o.lift_fy - o.armp_y * o.lift_fx;
    const float tx = o.tbx + (o.dragarm_tx + lax + o.drag_tx + o.am_tx);
    const float ty = o.tby + (o.dragarm_ty + lay + o.drag_ty + o.am_ty);
    const float tz = o.dragarm_tz + laz + o.drag_tz + o.am_tz;
    auto norm = [](float x, float y, float z) { return fast_sqrt(x * x + y * y + z * z); };
    const float sum_f = norm(o.drag_fx, o.drag_fy, o.fz_core) + norm(o.lift_fx, o.lift_fy, o.lift_fz) + norm(o.am_fx, o.am_fy, o.am_fz);
    const float sum_t = norm(o.tbx, o.tby, 0.0f) + norm(o.dragarm_tx, o.dragarm_ty, o.dragarm_tz) + norm(lax, lay, laz)
                      + norm(o.drag_tx, o.drag_ty, o.drag_tz) + norm(o.am_tx, o.am_ty, o.am_tz);
    return sum_f > kCancelGate * fmaxf(norm(fx, fy, fz), weight_scale) || sum_t > kCancelGate * fmaxf(norm(tx, ty, tz), weight_scale * lmax);
}

// fp64 reciprocal and square root for the fp64 re-evaluation: the fp32 hardware seeds (v_rcp_f32 / v_rsq_f32, 1 ulp)
// and two Newton steps in fp64 (2^-23 -> 2^-46 -> below fp64 resolution) - 7 and 11 instructions where the
// IEEE-exact sequences the compiler expands `/` and sqrt() to take ~15 and ~25.  Arguments are within fp32 range
// wherever the result is used (guarded by the model's own 1e-6 thresholds); sqrt64 returns 0 below 1e-30.
HYDRO_FN double rcp64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = (double)__builtin_amdgcn_rcpf((float)x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return r;
#else
    return 1.0 / x;
#endif
}
HYDRO_FN double sqrt64(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = (double)__builtin_amdgcn_rsqf((float)x), h = 0.5 * r;
    double y = x * r;
    y = __builtin_fma(__builtin_fma(-y, y, x), h, y);
    y = __builtin_fma(__builtin_fma(-y, y, x), h, y);
    return x > 1e-30 ? y : 0.0;
#else
    return x > 1e-30 ? sqrt(x) : 0.0;
#endif
}

// The whole wrench of one body in fp64, from the raw inputs: A13 (finite-difference acceleration), A1-A10 and
// A14-A15 as the reference's float64 Numba path evaluates them (numba_hydrodynamics.py:256-314,
// hydrodynamics_behavior.py:196-226), results rounded to fp32 once.  Called for the bodies assemble_wrench flags
// (a wavefront runs it when any of its lanes is flagged; each lane keeps the result only if ITS OWN flag is set,
// so what a body gets never depends on its neighbours).  Not on the fast path: plain fp64 sqrt and divisions.
// The 27 keypoint tests are taken from the fp32 pass (`wetmask`: they are comparisons of fp64-derived heights
// against zero and agree unless a keypoint is within 1e-7 edge lengths of the surface); every other branch of the
// model is re-decided here in fp64.
HYDRO_FN Wrench wrench_fp64(const BodyIn& b, uint32_t wetmask, float mass, double rho, double g, double inv_dt, bool warp, bool mine = true)
{
    // (thresholds as the reference's float64 literals: (double)0.2f is 1.5e-8 away from 0.2, and speed / 0.2 is arithmetic)
    const double qx = b.qx, qy = b.qy, qz = b.qz, qw = b.qw;
    const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const double xx = qx * x2, xy = qx * y2, xz = qx * z2, yy = qy * y2, yz = qy * z2, zz = qz * z2;
    const double sx = qw * x2, sy = qw * y2, sz = qw * z2;
    const double r00 = 1.0 - (yy + zz), r01 = xy - sz, r02 = xz + sy;           // A1 (:14-49), not normalised (N7)
    const double r10 = xy + sz, r11 = 1.0 - (xx + zz), r12 = yz - sx;
    const double r20 = xz - sy, r21 = yz + sx, r22 = 1.0 - (xx + yy);
    const double dx = b.dimx, dy = b.dimy, dz = b.dimz;
    const double hx = 0.5 * dx, hy = 0.5 * dy, hz = 0.5 * dz, vol = dx * dy * dz;
    const double pz = b.pz;
    // A3: extent, submersion ratio (:86-96)
    const double ex = hx * r20, ey = hy * r21, ez = hz * r22;
    const double extent = fabs(ex) + fabs(ey) + fabs(ez);
    const double zlo = pz - extent, zhi = pz + extent;
    const bool dry = zlo >= 0.0, full = zhi <= 0.0;
    const double height = zhi - zlo;
    double ratio = fmin(1.0, -zlo * rcp64(height));
    if (height < 1e-6) ratio = 1.0;
    if (full) ratio = 1.0;
    if (dry) ratio = 0.0;
    Wrench w;
    w.fx = w.fy = w.fz = w.tx = w.ty = w.tz = 0.0f; w.scale = 1.0f; w.ill = false;
    const bool wet = ratio > 1e-9;                                  // A4 (:277-279): zeros below (no early return: the
                                                                    // any_lane() votes further down need every lane)
    // centre of buoyancy as a body-frame lever arm: mean lattice index of the wet keypoints (:69-84,99-103)
    const int cnt = __builtin_popcount(wetmask);
    const int s_i = __builtin_popcount(wetmask & lattice_mask(0, 2)) - __builtin_popcount(wetmask & lattice_mask(0, 0));
    const int s_j = __builtin_popcount(wetmask & lattice_mask(1, 2)) - __builtin_popcount(wetmask & lattice_mask(1, 0));
    const int s_k = __builtin_popcount(wetmask & lattice_mask(2, 2)) - __builtin_popcount(wetmask & lattice_mask(2, 0));
    // (`mine`: this lane is one of the flagged ones.  The blocks a wavefront's flagged bodies do not need - the lattice
    // mean of a fully submerged body, the added mass of a body without added-mass coefficients - are skipped
    // wave-uniformly: the length of this pass is the tail of the launch.)
    double abx = 0.0, aby = 0.0, abz = 0.0;
    if (any_lane(mine && !dry && !full && cnt > 0)) {
        const double inv_cnt = (!dry && !full && cnt > 0) ? rcp64((double)cnt) : 0.0;
        const double lbx = hx * (double)s_i * inv_cnt, lby = hy * (double)s_j * inv_cnt, lbz = hz * (double)s_k * inv_cnt;
        abx = r00 * lbx + r01 * lby + r02 * lbz;
        aby = r10 * lbx + r11 * lby + r12 * lbz;
        abz = r20 * lbx + r21 * lby + r22 * lbz;
    }
    const double buoy = rho * (ratio * vol) * g;                                // A5 (:282)
    // A6 (:285-289)
    const double vx = b.vx, vy = b.vy, vz = b.vz;
    const double speed = sqrt64(vx * vx + vy * vy + vz * vz);
    const bool moving = speed > 1e-6;
    const double inv_speed = moving ? rcp64(speed) : 0.0;
    const double hx_ = vx * inv_speed, hy_ = vy * inv_speed, hz_ = vz * inv_speed;           // v_hat (0 at rest)
    // A7 (:108-143): u = R^T v_hat; the face of axis a that opposes the flow has sign s_a = -sign(u_a)
    const double ux = r00 * hx_ + r10 * hy_ + r20 * hz_;
    const double uy = r01 * hx_ + r11 * hy_ + r21 * hz_;
    const double uz = r02 * hx_ + r12 * hy_ + r22 * hz_;
    const double fsx = (ux < 0.0) ? 1.0 : -1.0, fsy = (uy < 0.0) ? 1.0 : -1.0, fsz = (uz < 0.0) ? 1.0 : -1.0;
    const double fax = ((ux != 0.0) && (pz + fsx * ex < 0.0)) ? fabs(ux) * (dy * dz) : 0.0;
    const double fay = ((uy != 0.0) && (pz + fsy * ey < 0.0)) ? fabs(uy) * (dx * dz) : 0.0;
    const double faz = ((uz != 0.0) && (pz + fsz * ez < 0.0)) ? fabs(uz) * (dx * dy) : 0.0;
    const double area = fax + fay + faz;                                        // 0 at rest: N1 completion
    const bool has_area = area > 1e-6;
    const double inv_area = has_area ? rcp64(area) : 0.0;
    const double lpx = fsx * hx * fax * inv_area, lpy = fsy * hy * fay * inv_area, lpz = fsz * hz * faz * inv_area;
    double apx = r00 * lpx + r01 * lpy + r02 * lpz;
    double apy = r10 * lpx + r11 * lpy + r12 * lpz;
    double apz = r20 * lpx + r21 * lpy + r22 * lpz;
    if (!has_area) { apx = abx; apy = aby; apz = abz; }                         // cop = cob (:115,140)
    // A8 (:146-182)
    const double half_rho = 0.5 * rho;
    const double lin_quad = moving ? half_rho * (speed * speed) * (double)b.cd_lin * area : 0.0;
    const double lin_scale = (speed < 0.2) ? speed * 5.0 : 1.0;
    const double damp_l = (double)b.damp_lin * lin_scale;
    const double fdx = -(lin_quad * hx_ + damp_l * vx) * ratio;
    const double fdy = -(lin_quad * hy_ + damp_l * vy) * ratio;
    const double fdz = -(lin_quad * hz_ + damp_l * vz) * ratio;
    const double ox = b.wx, oy = b.wy, oz = b.wz;
    const double wspeed = sqrt64(ox * ox + oy * oy + oz * oz);
    const bool spinning = wspeed > 1e-6;
    const double ang_quad = spinning ? half_rho * wspeed * (double)b.cd_ang * vol : 0.0;    // (1/2 rho w^2 Cd V) / w
    const double ang_scale = (wspeed < 0.2) ? wspeed * 5.0 : 1.0;
    const double ang_k = -(ang_quad + (double)b.damp_ang * ang_scale) * ratio;
    const double tdx = ang_k * ox, tdy = ang_k * oy, tdz = ang_k * oz;
    // A9 (:185-217)
    double flx = 0.0, fly = 0.0, flz = 0.0;
    if (!(speed < 1e-6)) {
        const double d = fmin(1.0, fmax(-1.0, -uz));                            // -up . v_hat, up = R[:,2]
        const double c_l = 2.0 * d * sqrt64(fmax(0.0, (1.0 - d) * (1.0 + d)));    // sin(2 asin d)
        const double lift = half_rho * (speed * speed) * c_l * area * (double)b.lift;
        const double axx = hy_ * r22 - hz_ * r12, axy = hz_ * r02 - hx_ * r22, axz = hx_ * r12 - hy_ * r02;   // v_hat x up
        const double n_axis = sqrt64(axx * axx + axy * axy + axz * axz);
        if (!(n_axis < 1e-6)) {
            const double k = lift * ratio * rcp64(n_axis);
            flx = k * (axy * hz_ - axz * hy_); fly = k * (axz * hx_ - axx * hz_); flz = k * (axx * hy_ - axy * hx_);
        }
    }
    // A13 + A10 (hydrodynamics_behavior.py:196-202; numba_hydrodynamics.py:220-253)
    double fax_ = 0.0, fay_ = 0.0, faz_ = 0.0, tax = 0.0, tay = 0.0, taz = 0.0;
    if (any_lane(mine && (b.am_lin != 0.0f || b.am_ang != 0.0f))) {
        const double ax = (vx - (double)b.pvx) * inv_dt, ay = (vy - (double)b.pvy) * inv_dt, az = (vz - (double)b.pvz) * inv_dt;
        const double bx = (ox - (double)b.pwx) * inv_dt, by = (oy - (double)b.pwy) * inv_dt, bz = (oz - (double)b.pwz) * inv_dt;
        double alx, aly, alz, blx, bly, blz;                                    // accelerations in the "local" frame
        if (warp) {                                                             // N3: quat_rotate = R (warp_hydrodynamics.py:216-217)
            alx = r00 * ax + r01 * ay + r02 * az; aly = r10 * ax + r11 * ay + r12 * az; alz = r20 * ax + r21 * ay + r22 * az;
            blx = r00 * bx + r01 * by + r02 * bz; bly = r10 * bx + r11 * by + r12 * bz; blz = r20 * bx + r21 * by + r22 * bz;
        } else {                                                                // R^T (numba_hydrodynamics.py:229-230)
            alx = r00 * ax + r10 * ay + r20 * az; aly = r01 * ax + r11 * ay + r21 * az; alz = r02 * ax + r12 * ay + r22 * az;
            blx = r00 * bx + r10 * by + r20 * bz; bly = r01 * bx + r11 * by + r21 * bz; blz = r02 * bx + r12 * by + r22 * bz;
        }
        const double rv = vol * rho;
        const double kf = -(rv * (double)b.am_lin) * ratio, kt = -(rv * (double)b.am_ang) * ratio;
        const double glx = kf * alx, gly = kf * aly, glz = kf * alz;
        const double tlx = kt * (dy * dy + dz * dz) * blx, tly = kt * (dx * dx + dz * dz) * bly, tlz = kt * (dx * dx + dy * dy) * blz;
        fax_ = r00 * glx + r01 * gly + r02 * glz; fay_ = r10 * glx + r11 * gly + r12 * glz; faz_ = r20 * glx + r21 * gly + r22 * glz;
        tax = r00 * tlx + r01 * tly + r02 * tlz; tay = r10 * tlx + r11 * tly + r12 * tlz; taz = r20 * tlx + r21 * tly + r22 * tlz;
    }
    // A14 (hydrodynamics_behavior.py:212-218)
    const double gx = fdx + flx, gy = fdy + fly, gz = fdz + flz;                // drag + lift act at the centre of pressure
    const double fx = gx + fax_, fy = gy + fay_, fz = buoy + (gz + faz_);
    const double tx = aby * buoy + (apy * gz - apz * gy) + tdx + tax;
    const double ty = -abx * buoy + (apz * gx - apx * gz) + tdy + tay;
    const double tz = (apx * gy - apy * gx) + tdz + taz;
    // A15 (:220-226)
    const double scale = fmin(1.0, (double)mass * 500.0 * rcp64(sqrt64(fx * fx + fy * fy + fz * fz) + 1e-6));
    if (wet) {
        w.fx = (float)(fx * scale); w.fy = (float)(fy * scale); w.fz = (float)(fz * scale);
        w.tx = (float)(tx * scale); w.ty = (float)(ty * scale); w.tz = (float)(tz * scale);
        w.scale = (float)scale;
    }
    return w;
}

// One body of the fused entry points (A13 is done by the caller in fp32 for the fast pass: b.ax..b.bz; the raw
// previous velocity travels along for the fp64 pass).  k_lin / k_ang: clamped drag coefficients for the implicit
// integrator (drag_force = k_lin v, drag_torque = k_ang w), from the fp32 pass in both cases.
// `reload(mass)` hands the body's inputs over a second time for the fp64 pass.  The kernels RE-READ them from
// memory there instead of keeping 29 input registers alive across the whole fp32 pass (that costs a wave per
// SIMD: 128+ VGPRs); the pass runs for one wavefront in ~50, so the extra traffic is ~2 % of lines that are
// still in L2.  The host instantiation just returns the struct it has.
template <typename Reload>
HYDRO_FN Wrench solve_wrench(const BodyIn& b, float mass, double rho, double g, double inv_dt, bool warp, Reload reload,
                             float* k_lin = nullptr, float* k_ang = nullptr, float* sub_ratio = nullptr)
{
    const BodyOut o = solve_body<false>(b, rho, g, warp);
    if (sub_ratio) *sub_ratio = o.wet ? o.ratio : 0.0f;
    const float weight_scale = kCancelFloor * ((float)(rho * g) * (b.dimx * b.dimy * b.dimz));
    const float lmax = fmaxf(b.dimx, fmaxf(b.dimy, b.dimz));
    Wrench w = assemble_wrench(o, mass, weight_scale, lmax);
    if (k_lin) *k_lin = o.wet ? o.lin_k * w.scale : 0.0f;
    if (k_ang) *k_ang = o.wet ? o.ang_k * w.scale : 0.0f;
    const uint32_t wetmask = o.wetmask;
    if (__builtin_expect(any_lane(w.ill), 0)) {             // wave-uniform branches, cold
        raise_priority();                                   // the flagged wavefronts are the tail of the launch
        w.ill = w.ill && cancels_2norm(o, weight_scale, lmax);
        if (any_lane(w.ill)) {
            float mass2;
            const BodyIn b2 = reload(mass2);
            const Wrench r = wrench_fp64(b2, wetmask, mass2, rho, g, inv_dt, warp, w.ill);
            if (w.ill) { w.fx = r.fx; w.fy = r.fy; w.fz = r.fz; w.tx = r.tx; w.ty = r.ty; w.tz = r.tz; w.scale = r.scale; }
        }
    }
    return w;
}

}  // namespace hydro
